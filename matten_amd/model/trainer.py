"""
A minimal stand-in for ``pytorch_lightning.Trainer`` with the call sequence of ``Trainer.fit`` / ``Trainer.test`` that
the reference's training script relies on (scripts/train_materials_tensor.py:52-66): configure_optimizers, then per
epoch training_step / backward / step over the train loader, validation_step over the val loader, the epoch-end hooks,
and a ReduceLROnPlateau-style scheduler stepped on the monitored score.  For the boxes without Lightning and for tests;
with Lightning installed use its Trainer (the model is a LightningModule then).
"""
from typing import Optional

import torch


class Trainer:
    def __init__(self, max_epochs: int = 1, limit_train_batches: Optional[int] = None, accelerator: str = "gpu",
                 devices: int = 1, **ignored):
        self.max_epochs, self.limit_train_batches = max_epochs, limit_train_batches
        self.history = []

    def fit(self, model, datamodule=None, train_dataloaders=None, val_dataloaders=None):
        train = train_dataloaders if train_dataloaders is not None else datamodule.train_dataloader()
        val = val_dataloaders if val_dataloaders is not None else (datamodule.val_dataloader() if datamodule else None)
        cfg = model.configure_optimizers()
        optimizer = cfg["optimizer"] if isinstance(cfg, dict) else cfg
        scheduler = cfg.get("lr_scheduler") if isinstance(cfg, dict) else None
        for epoch in range(self.max_epochs):
            model.train()
            for i, batch in enumerate(train):
                if self.limit_train_batches is not None and i >= self.limit_train_batches:
                    break
                loss = model.training_step(batch, i)["loss"]
                optimizer.zero_grad(set_to_none=True)
                loss.backward()
                optimizer.step()
            model.on_training_epoch_end()
            if val is not None:
                model.eval()
                with torch.no_grad():
                    for i, batch in enumerate(val):
                        model.validation_step(batch, i)
                model.on_validation_epoch_end()
            score = model.logged.get(model.monitor_key) if hasattr(model, "logged") else None
            if scheduler is not None:
                if isinstance(scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau):
                    if score is not None:
                        scheduler.step(score)
                else:
                    scheduler.step()
            self.history.append({"epoch": epoch, **{k: (float(v) if hasattr(v, "item") else v)
                                                    for k, v in getattr(model, "logged", {}).items()}})
        return self

    def test(self, model=None, datamodule=None, dataloaders=None, ckpt_path=None):
        loader = dataloaders if dataloaders is not None else datamodule.test_dataloader()
        model.eval()
        with torch.no_grad():
            for i, batch in enumerate(loader):
                model.test_step(batch, i)
        model.on_test_epoch_end()
        return [{k: (float(v) if hasattr(v, "item") else v) for k, v in getattr(model, "logged", {}).items()
                 if k.startswith(("test/", "metric_test/"))}]
