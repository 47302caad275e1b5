"""`main.py analyze` -- batch-size-1 evaluation over the dev split, then the train split (drop-in for
src/cli/analyze.py:49-242): same flags/defaults (:23-47; --predict-grf-components [1] only), loads the
latest checkpoint from <checkpoint-dir>/<model-type>/, runs model forward + loss evaluator under no_grad,
appends one row per window to dev_analysis.csv / train_analysis.csv, prints the averaged report.
Forward and loss run on the HIP kernels (batch 1 is launch-latency bound; no backward here)."""
import argparse
import csv
import logging
import os

import torch
from torch.utils.data import DataLoader

from ..loss.RegressionLossEvaluator import RegressionLossEvaluator
from ._common import add_additive_flags, add_component_flags, dtype_of, open_dataset, pick_device
from .abstract_command import AbstractCommand


class AnalyzeCommand(AbstractCommand):
    def __init__(self):
        super().__init__()

    def register_subcommand(self, subparsers: argparse._SubParsersAction):
        p = subparsers.add_parser('analyze', help='Evaluate the performance of a model on dataset.')
        p.add_argument('--dataset-home', type=str, default='../data', help='The path to the AddBiomechanics dataset.')
        p.add_argument('--model-type', type=str, default='feedforward', help='The model to evaluate.')
        p.add_argument('--no-wandb', action='store_true', default=False, help='Do not log to Weights and Biases.')
        p.add_argument('--output-data-format', type=str, default='all_frames', choices=['all_frames', 'last_frame'])
        p.add_argument('--checkpoint-dir', type=str, default='../checkpoints')
        p.add_argument('--geometry-folder', type=str, default=None)
        p.add_argument('--history-len', type=int, default=50)
        p.add_argument('--stride', type=int, default=5)
        p.add_argument('--hidden-dims', type=int, nargs='+', default=[512, 512])
        p.add_argument('--activation', type=str, default='sigmoid')
        p.add_argument('--short', type=bool, default=False)
        p.add_argument('--data-loading-workers', type=int, default=3)
        add_component_flags(p, train_defaults=False)
        add_additive_flags(p)
        p.add_argument('--max-windows', type=int, default=0, help='Stop each split after this many windows (0 = all).')

    def run(self, args: argparse.Namespace):
        if 'command' in args and args.command != 'analyze':
            return False
        checkpoint_dir = os.path.join(os.path.abspath(args.checkpoint_dir), args.model_type)
        os.makedirs(checkpoint_dir, exist_ok=True)
        device = pick_device(args)
        geometry = self.ensure_geometry(args.geometry_folder)
        model = None
        for split, csv_name in (('dev', 'dev_analysis.csv'), ('train', 'train_analysis.csv')):
            logging.info(f'## Loading {split} dataset:')
            dataset = open_dataset(args, split, args.history_len, args.stride, args.output_data_format, geometry)
            if model is None:
                model = self.get_model(dataset.num_dofs, dataset.num_contact_bodies, args.model_type,
                                       history_len=args.history_len, hidden_dims=args.hidden_dims,
                                       activation=args.activation, stride=args.stride, batchnorm=False, dropout=False,
                                       dropout_prob=0.0, root_history_len=10,
                                       output_data_format=args.output_data_format, device=device,
                                       compute_dtype=dtype_of(args)).to(device)
                self.load_latest_checkpoint(model, checkpoint_dir=checkpoint_dir)
                model.eval()
            evaluator = RegressionLossEvaluator(dataset=dataset, split=split, device=device)
            loader = DataLoader(dataset, batch_size=1, shuffle=False, num_workers=args.data_loading_workers)
            compute_report = bool(getattr(dataset, 'skeletons', None))   # inverse dynamics needs nimble skeletons
            n = len(loader)
            with torch.no_grad(), open(os.path.join(checkpoint_dir, csv_name), 'a') as f:
                writer = None
                for i, (inputs, labels, subj, trial) in enumerate(loader):
                    outputs = model(inputs)
                    evaluator(inputs, outputs, labels, subj, trial, args, compute_report=compute_report)
                    stats = {"sub_name": window_subject(dataset, subj), "trial_name": window_trial(dataset, subj, trial)}
                    writer = writer or csv.DictWriter(f, fieldnames=stats.keys())
                    writer.writerow(stats)
                    last = (i == n - 1) or (args.max_windows and i + 1 >= args.max_windows)
                    if (i + 1) % 100 == 0 or last:
                        logging.info(f'  - Batch {i + 1}/{n}')
                    if (i + 1) % 1000 == 0 or last:
                        evaluator.print_report(args, reset=False, log_to_wandb=not args.no_wandb)
                    if last:
                        break
            print(f'Final {split} results:')
            evaluator.print_report(log_to_wandb=False)
        return True


def window_subject(dataset, subj) -> str:
    i = int(subj[0]) if len(subj) else 0
    if hasattr(dataset, 'subject_paths'):
        return os.path.basename(dataset.subject_paths[i])
    return f"synthetic_subject_{i}"


def window_trial(dataset, subj, trial) -> str:
    j = int(trial[0]) if len(trial) else 0
    if hasattr(dataset, 'subjects'):
        return str(dataset.subjects[int(subj[0])].getTrialName(j))
    return f"window_{j}"
