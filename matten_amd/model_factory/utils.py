"""Chain dict-passing modules, threading irreps_out -> irreps_in (mirrors reference model_factory/utils.py:13-91)."""
from collections import OrderedDict
from typing import Dict, Optional, Tuple

from ..nn.sequential import Sequential
from ..o3 import Irreps


def create_sequential_module(
    modules: "OrderedDict[str, Tuple[type, Dict]]",
    irreps_in: Optional[Dict[str, Irreps]] = None,
    use_kwargs_irreps_in: bool = False,
) -> Sequential:
    built = OrderedDict()
    prev = None
    for name, (cls_type, kwargs) in modules.items():
        ir = irreps_in if prev is None else prev.irreps_out
        if "irreps_in" in kwargs:
            if not use_kwargs_irreps_in:
                raise ValueError(
                    f"Trying to automatically determine irreps_in for module {name} "
                    f"But it is provided as kwargs. Set `use_kwargs_irrpes_in=True` to force it."
                )
            ir = dict(ir or {})
            ir.update(kwargs["irreps_in"])
        kw = dict(kwargs)
        kw["irreps_in"] = ir
        try:
            prev = cls_type(**kw)
        except Exception as e:
            raise RuntimeError(f"Failed instantiate module `{cls_type.__name__}` with kwargs: `{kw}`") from e
        built[name] = prev
        # anomaly detection behind every layer in debug mode (reference model_factory/utils.py:82-87)
        from ..log import get_log_level

        if get_log_level() == "DEBUG":
            from ..nn.utils import DetectAnomaly

            built[DetectAnomaly.__name__ + "_" + name] = DetectAnomaly(irreps_in=prev.irreps_out, name=name)
    return Sequential(built)


class UnsupportedConfig(NotImplementedError):
    """a hyper-parameter combination of the reference that the gfx950 kernels do not cover"""


def validate_hparams(hparams: Dict, dataset_hparams: Dict = None) -> None:
    """One up-front check of the backbone hyper-parameters against what the accelerated path implements, so that an
    unsupported reference config fails at construction with the full list instead of deep inside a layer.

    Supported envelope (covers every config the reference ships: pretrained/20230627/config_final.yaml,
    scripts/configs/{materials_tensor,atomic_tensor}.yaml, tests/model/test_tfn_tensor.py):
      irreps_edge_sh            0e+1o+...+lmax with lmax <= 4, parity (-1)^l
      radial_basis_type         bessel                          (reference nn/embedding.py:189-199)
      invariant_layers/neurons  2 x 32  (radial MLP [nb,32,32,W]; nb <= 16)   (nn/utils.py:246-251)
      nonlinearity_type         gate | norm                     (nn/utils.py:96-150)
      normalization             batch | instance | none         (nn/utils.py:414-418, 448-588)
      reduce                    mean | sum | min | max          (nn/nodewise.py:131,142-148)
      use_atom_feats            false | true (data['atom_feats'] [n_atoms, atom_feats_size])   (nn/embedding.py:59-68,103-105)
      dataset_hparams           allowed_species given
    """
    from ..o3 import Irreps

    problems = []
    try:
        sh = Irreps(hparams["irreps_edge_sh"])
        lmax = len(sh) - 1
        if lmax > 4 or sh != Irreps.spherical_harmonics(lmax):
            problems.append(f"irreps_edge_sh={hparams['irreps_edge_sh']!r}: must be 0e+1o+...+lmax with lmax <= 4")
    except Exception as e:  # noqa: BLE001
        problems.append(f"irreps_edge_sh={hparams.get('irreps_edge_sh')!r}: {e}")
    if str(hparams.get("radial_basis_type", "bessel")).lower() != "bessel":
        problems.append(f"radial_basis_type={hparams['radial_basis_type']!r}: only 'bessel'")
    if int(hparams.get("num_radial_basis", 8)) > 16:
        problems.append(f"num_radial_basis={hparams['num_radial_basis']}: at most 16")
    if int(hparams.get("invariant_layers", 2)) != 2 or int(hparams.get("invariant_neurons", 32)) != 32:
        problems.append(f"invariant_layers={hparams.get('invariant_layers')}, invariant_neurons="
                        f"{hparams.get('invariant_neurons')}: the radial MLP is fixed at 2 hidden layers of 32")
    if str(hparams.get("nonlinearity_type", "gate")).lower() not in ("gate", "norm"):
        problems.append(f"nonlinearity_type={hparams['nonlinearity_type']!r}: only 'gate' or 'norm'")
    norm = hparams.get("normalization")
    if norm is not None and str(norm).lower() not in ("batch", "instance", "none"):
        problems.append(f"normalization={norm!r}: only 'batch', 'instance' or none")
    if str(hparams.get("reduce", "mean")).lower() not in ("mean", "sum", "min", "max"):
        problems.append(f"reduce={hparams['reduce']!r}: one of 'mean', 'sum', 'min', 'max' (nn/nodewise.py:131)")
    if hparams.get("use_atom_feats", False) and dataset_hparams is not None and not dataset_hparams.get("atom_feats_size"):
        problems.append("use_atom_feats=True needs dataset_hparams['atom_feats_size'] (reference nn/embedding.py:60-64)")
    if dataset_hparams is not None and not dataset_hparams.get("allowed_species"):
        problems.append("dataset_hparams['allowed_species'] is required")
    if problems:
        raise UnsupportedConfig(
            "this backbone configuration is valid for the reference but outside what matten_amd's MI355X kernels "
            "implement:\n  - " + "\n  - ".join(problems) + "\n(see validate_hparams.__doc__ for the supported envelope)"
        )
