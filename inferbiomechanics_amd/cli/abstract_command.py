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


def flat_to_torch_optimizer_state(sd, model):
    """HipTrainer.optimizer_state_dict() payload -> torch.optim grammar (parameters in model.parameters() order), so a
    checkpoint of the fused trainer resumes under `train --eager` (torch.optim over the drop-in modules)"""
    from ..engine import torch_param_groups, torch_state_from_flat
    shapes = {k: p.shape for k, p in model.named_parameters()}
    names = list(shapes)
    state = torch_state_from_flat(sd['opt_type'], int(sd['step']), names, lambda k: shapes[k], sd['layout'], sd['s1'], sd['s2'])
    return {'state': state, 'param_groups': torch_param_groups(sd['opt_type'], sd['lr'], len(names))}


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
        dev = self._model_device(device)
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

    @staticmethod
    def _model_device(device):
        """The reference default is device='cpu' (and analyze / visualize default to --device cpu).  This build has no CPU
        path, so 'cpu' resolves to the local MI355X (with a log line); without a GPU the construction fails HERE, naming
        the argument to pass, instead of building a module whose first call raises."""
        from .. import hip
        if device != 'cpu':
            return resolve_device(device)
        if hip._dry_run:                                   # tests/test_plumbing_cpu.py only
            return 'cpu'
        if torch.cuda.is_available():
            logging.info("get_model(device='cpu'): this build runs on the GPU only -- using the local MI355X")
            return resolve_device('gpu')
        raise hip.HipError("get_model(device='cpu'): the HIP path has no CPU fallback and no GPU is visible; "
                           "pass device='gpu' (CLI: --device gpu) on an MI355X box")

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
        osd = checkpoint.get('optimizer_state_dict')
        if optimizer is not None and hasattr(optimizer, 'load_optimizer_state_dict'):      # HipTrainer
            if osd is not None:
                optimizer.load_optimizer_state_dict(osd)     # its own flat payload, or a torch.optim state dict
            optimizer.refresh_after_param_load()             # always: the parameters changed under the bf16 shadow
        elif optimizer is not None and osd is not None:
            if 'layout' in osd and 'state' not in osd:       # written by the fused trainer, loaded under --eager
                osd = flat_to_torch_optimizer_state(osd, target)
            optimizer.load_state_dict(osd)
        epoch = checkpoint['epoch']
        batch = key(checkpoints[-1])[1]
        print(f"Loaded checkpoint from epoch {epoch}, batch {batch}")
        return epoch, int(batch)
