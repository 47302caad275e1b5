"""Flag groups and factories shared by the train / analyze / visualize commands.  Flag names and defaults
are the reference's (src/cli/train.py:24-69, analyze.py:23-47, visualize.py:21-56); the additive flags
(`--device` on train, `--synthetic-windows`, `--compute-dtype`, `--max-steps`, `--eager`) are new."""
import argparse
import logging
import os
from typing import Optional

import torch

from ..data.AddBiomechanicsDataset import (AddBiomechanicsDataset, MotionWindowView, SyntheticMotionWindows,
                                               SyntheticWindowDataset)


def add_component_flags(p: argparse.ArgumentParser, train_defaults: bool):
    d6 = list(range(6)) if train_defaults else None
    p.add_argument('--predict-grf-components', type=int, nargs='+', default=d6 if train_defaults else [1],
                   help='Which grf components to train.')
    p.add_argument('--predict-cop-components', type=int, nargs='*', default=d6 if train_defaults else [],
                   help='Which cop components to train.')
    p.add_argument('--predict-moment-components', type=int, nargs='*', default=d6 if train_defaults else [],
                   help='Which moment components to train.')
    p.add_argument('--predict-wrench-components', type=int, nargs='*',
                   default=list(range(12)) if train_defaults else [], help='Which wrench components to train.')


def add_additive_flags(p: argparse.ArgumentParser, device_default: str = 'gpu'):
    p.add_argument('--device', type=str, default=device_default,
                   help="Where to run: 'gpu' (HIP kernels on the local MI355X). 'cpu' is refused: no CPU path.")
    p.add_argument('--synthetic-windows', type=int, default=0,
                   help='Use N seeded synthetic windows per split instead of .b3d files (no nimblephysics needed).')
    p.add_argument('--compute-dtype', type=str, default='fp32', choices=['fp32', 'bf16'],
                   help='Storage type of activations/weights in the kernels (fp32 = parity mode).')
    p.add_argument('--feat-dim', type=int, default=300, help='[diffusion models] features per frame.')


def dtype_of(args) -> torch.dtype:
    return torch.bfloat16 if getattr(args, 'compute_dtype', 'fp32') == 'bf16' else torch.float32


def is_diffusion(model_type: str) -> bool:
    return model_type.startswith('diffusion')


def open_dataset(args, split: str, history_len: int, stride: int, output_data_format: str, geometry: Optional[str]):
    """`.b3d` windows through data/AddBiomechanicsDataset.py (needs the nimblephysics wheel at run time), or seeded
    synthetic windows with the same per-item layout.  The diffusion models see a window as one [F, D] matrix
    (`MotionWindowView`: the model-input channels and the four label blocks of every frame side by side)."""
    n = getattr(args, 'synthetic_windows', 0)
    model_type = getattr(args, 'model_type', 'feedforward')
    if n > 0:
        seed = {'train': 0, 'dev': 1, 'test': 2}.get(split, 3)
        if is_diffusion(model_type):
            return SyntheticMotionWindows(n, window=history_len // stride if stride > 1 else history_len,
                                          feat=args.feat_dim, seed=seed)
        return SyntheticWindowDataset(n, history_len, stride, output_data_format=output_data_format, seed=seed,
                                      history_width=30 if model_type == 'groundlink' else 0)   # root_history_len = 10
    path = os.path.abspath(os.path.join(args.dataset_home, split))
    try:
        from ..data.AddBiomechanicsDataset import _nimble
        _nimble()
    except ImportError:
        raise SystemExit(f"Reading {path} needs the `nimblephysics` .b3d loader, which is not installed. "
                         f"Pass --synthetic-windows N to run the hot path on seeded synthetic windows.")
    # the reference call (train.py:141-148, analyze.py:82-88, visualize.py:94-101)
    dataset = AddBiomechanicsDataset(path, history_len, geometry, device=torch.device('cpu'), stride=stride,
                                     output_data_format=output_data_format,
                                     testing_with_short_dataset=bool(getattr(args, 'short', False)),
                                     skip_loading_skeletons=not getattr(args, 'compute_report', False)
                                     and getattr(args, 'command', 'train') == 'train')
    if is_diffusion(model_type):
        return MotionWindowView(dataset)
    return dataset


def pick_device(args) -> torch.device:
    from .. import hip
    if hip._dry_run:                      # tests/test_plumbing_cpu.py only
        return torch.device('cpu')
    if str(args.device) == 'cpu':
        raise SystemExit("--device cpu: this build has no CPU path (the reference's PyTorch-CPU path is the "
                         "reference itself). Use --device gpu.")
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: the HIP path needs an MI355X")
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('IB_BENCH_REHEARSAL') == '1':      # several ranks on a one-GPU box (control-flow rehearsal over gloo)
        local = 0
    torch.cuda.set_device(local)
    return torch.device('cuda', local)
