"""Dataset side of the training script (mirror of the reference's matten.dataset package): see structure_scalar_tensor."""
