"""`main.py visualize` -- flag surface of src/cli/visualize.py:21-56 kept; the body of the reference command
is a nimblephysics NimbleGUI browser playback (:126-258), which is third-party GUI code outside the GPU hot
path.  What IS on the path -- a batch-1 model forward + loss evaluation per tick
(:157-200) -- is run here over the first windows of the split and printed, so a checkpoint can be inspected
without the GUI."""
import argparse
import os

import torch

from ..loss.RegressionLossEvaluator import RegressionLossEvaluator
from ._common import add_additive_flags, add_component_flags, dtype_of, open_dataset, pick_device
from .abstract_command import AbstractCommand


class VisualizeCommand(AbstractCommand):
    def __init__(self):
        super().__init__()

    def register_subcommand(self, subparsers: argparse._SubParsersAction):
        p = subparsers.add_parser('visualize', help='Visualize the performance of a model on dataset.')
        p.add_argument('--dataset-home', type=str, default='../data')
        p.add_argument('--model-type', type=str, default='feedforward')
        p.add_argument('--output-data-format', type=str, default='all_frames', choices=['all_frames', 'last_frame'])
        p.add_argument('--checkpoint-dir', type=str, default='../checkpoints')
        p.add_argument('--geometry-folder', type=str, default=None)
        p.add_argument('--history-len', type=int, default=50)
        p.add_argument('--stride', type=int, default=5)
        p.add_argument('--dropout', action='store_true')
        p.add_argument('--dropout-prob', type=float, default=0.5)
        p.add_argument('--hidden-dims', type=int, nargs='+', default=[512, 512])
        p.add_argument('--batchnorm', action='store_true')
        p.add_argument('--activation', type=str, default='sigmoid')
        p.add_argument('--batch-size', type=int, default=32)
        p.add_argument('--short', action='store_true')
        add_component_flags(p, train_defaults=True)
        add_additive_flags(p)
        p.add_argument('--num-frames', type=int, default=8, help='How many windows to evaluate and print.')

    def run(self, args: argparse.Namespace):
        if 'command' in args and args.command != 'visualize':
            return False
        try:
            import nimblephysics  # noqa: F401
            have_gui = True
        except ImportError:
            have_gui = False
        checkpoint_dir = os.path.join(os.path.abspath(args.checkpoint_dir), args.model_type)
        device = pick_device(args)
        dataset = open_dataset(args, 'test', args.history_len, args.stride, args.output_data_format,
                               self.ensure_geometry(args.geometry_folder))
        model = self.get_model(dataset.num_dofs, dataset.num_contact_bodies, args.model_type,
                               history_len=args.history_len, stride=args.stride, hidden_dims=args.hidden_dims,
                               activation=args.activation, batchnorm=args.batchnorm, dropout=args.dropout,
                               dropout_prob=args.dropout_prob, root_history_len=10,
                               output_data_format=args.output_data_format, device=device,
                               compute_dtype=dtype_of(args)).to(device)
        self.load_latest_checkpoint(model, checkpoint_dir=checkpoint_dir)
        model.eval()
        evaluator = RegressionLossEvaluator(dataset=dataset, split='test', device=device)
        if not have_gui:
            print("nimblephysics is not installed: the NimbleGUI playback is unavailable; printing per-window "
                  "predictions vs labels instead.")
        with torch.no_grad():
            for frame in range(min(args.num_frames, len(dataset))):
                inputs, labels, subj, trial = dataset[frame]
                inputs = {k: v.unsqueeze(0) for k, v in inputs.items()}
                labels = {k: v.unsqueeze(0) for k, v in labels.items()}
                outputs = model(inputs)
                loss = evaluator(inputs, outputs, labels, [subj], [trial], args)
                print(f"window {frame}: loss {float(loss):.6f}")
        evaluator.print_report(args)
        return True
