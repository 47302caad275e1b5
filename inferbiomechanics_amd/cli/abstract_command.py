"""Command base class: subcommand interface, the MODEL REGISTRY and the checkpoint loader.

Drop-in for src/cli/abstract_command.py:11-120.  ``get_model`` keeps the reference signature and registry
names (``feedforward`` / ``groundlink`` / ``analytical``) and adds the two diffusion denoisers additively
(``diffusion-mlp`` / ``diffusion-transformer``).  ``load_latest_checkpoint`` keeps the file grammar
``epoch_{E}_batch_{i}.pt`` and the ordering rule (:92-103) and additionally accepts checkpoints whose keys
carry DDP's ``module.`` prefix (the reference saves the DDP wrapper, train.py:276, but loads into a bare
model, :110)."""
import argparse
import logging
import os
from typing import List

import torch

MODEL_TYPES = ['analytical', 'feedforward', 'groundlink', 'diffusion-mlp', 'diffusion-transformer']


def resolve_device(device) -> torch.device:
    """'gpu' / 'cuda' / int rank -> torch.device('cuda', i).  'cpu' is refused: the hot path is HIP-only."""
    if isinstance(device, torch.device):
        return device
    if isinstance(device, int):
        return torch.device('cuda', device)
    d = str(device)
    if d in ('gpu', 'cuda'):
        return torch.device('cuda', torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return torch.device(d)


class AbstractCommand:
    def register_subcommand(self, subparsers: argparse._SubParsersAction):
        pass

    def run(self, args: argparse.Namespace) -> bool:
        pass

    def register_model_options(self, parser: argparse.ArgumentParser):
        pass

    def ensure_geometry(self, geometry: str):
        """The reference downloads Geometry.zip with wget when missing (abstract_command.py:25-42); the bone
        meshes only feed the nimblephysics GUI / skeleton loader, which the hot path never touches, so no
        download is attempted here."""
        if geometry is None:
            geometry = './Geometry'
        geometry = os.path.abspath(geometry)
        return geometry if geometry.endswith('/') else geometry + '/'

    def get_model(self, num_dofs: int, num_contact_bodies: int, model_type: str = 'feedforward',
                  history_len: int = 5, stride: int = 1, hidden_dims: List[int] = [512], activation: str = 'relu',
                  batchnorm: bool = False, dropout: bool = False, dropout_prob: float = 0.0,
                  root_history_len: int = 10, output_data_format: str = 'all_frames', device: str = 'cpu',
                  compute_dtype: torch.dtype = torch.float32, feat_dim: int = 300, window: int = 50,
                  d_model: int = 512, num_heads: int = 8, dim_feedforward: int = 2048, num_layers: int = 4):
        dev = resolve_device(device) if device != 'cpu' else 'cpu'
        if model_type == 'feedforward':
            from ..models.FeedForwardRegressionBaseline import FeedForwardBaseline
            return FeedForwardBaseline(num_dofs, num_contact_bodies, history_len, output_data_format, activation,
                                       stride=stride, hidden_dims=hidden_dims, batchnorm=batchnorm, dropout=dropout,
                                       dropout_prob=dropout_prob, root_history_len=root_history_len, device=dev,
                                       compute_dtype=compute_dtype)
        if model_type == 'diffusion-mlp':
            from ..models.DiffusionDenoisers import DiffusionMLP
            return DiffusionMLP(feat_dim, hidden_dims, device=dev, compute_dtype=compute_dtype)
        if model_type == 'diffusion-transformer':
            from ..models.DiffusionDenoisers import DiffusionTransformer
            return DiffusionTransformer(feat_dim, window, d_model=d_model, num_heads=num_heads,
                                        dim_feedforward=dim_feedforward, num_layers=num_layers, device=dev,
                                        compute_dtype=compute_dtype)
        if model_type == 'groundlink':
            # the reference registry call raises TypeError (abstract_command.py:74-79 passes the feedforward argument
            # list to Groundlink.__init__, Groundlink.py:20); here the constructor gets its own arguments.  12 joint
            # centres per frame (AddBiomechanicsDataset.py:221-223), root history of `root_history_len` frames.
            from ..models.Groundlink import Groundlink
            return Groundlink(num_dofs, 12, root_history_len, output_data_format=output_data_format,
                              fc_dropout=dropout_prob if dropout else 0.2, device=dev, compute_dtype=compute_dtype)
        assert (model_type == 'analytical')
        raise NotImplementedError("model type 'analytical' is a nimblephysics CPU heuristic with no parameters "
                                  "(src/models/AnalyticalBaseline.py); it is outside the GPU hot path")

    def load_latest_checkpoint(self, model, optimizer=None, checkpoint_dir="../checkpoints"):
        if not os.path.exists(checkpoint_dir):
            print("Checkpoint directory does not exist!")
            return -1, 0
        checkpoints = [f for f in os.listdir(checkpoint_dir) if f.endswith(".pt")]
        if not checkpoints:
            print("No checkpoints available!")
            return -1, 0
        key = lambda name: (int(name.split('_')[1]), int(name.split('_')[3].split('.')[0]))
        checkpoints.sort(key=key)
        latest = os.path.join(checkpoint_dir, checkpoints[-1])
        logging.info(f"latest_checkpoint={latest!r}")
        checkpoint = torch.load(latest, map_location='cpu')
        state = {(k[len('module.'):] if k.startswith('module.') else k): v
                 for k, v in checkpoint['model_state_dict'].items()}
        target = model.module if hasattr(model, 'module') else model
        target.load_state_dict(state)
        if optimizer is not None and checkpoint.get('optimizer_state_dict') is not None:
            if hasattr(optimizer, 'load_optimizer_state_dict'):      # HipTrainer
                optimizer.load_optimizer_state_dict(checkpoint['optimizer_state_dict'])
                optimizer.refresh_after_param_load()
            else:
                optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
        epoch = checkpoint['epoch']
        batch = key(checkpoints[-1])[1]
        print(f"Loaded checkpoint from epoch {epoch}, batch {batch}")
        return epoch, int(batch)
