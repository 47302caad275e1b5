"""`main.py train` -- the DDP training driver (drop-in for src/cli/train.py).

Flags and defaults are the reference's (train.py:24-69).  Control flow kept: per epoch a no-grad dev
evaluation, a barrier, then the training loop with a report every 1000 batches and a rank-0 checkpoint
``<checkpoint-dir>/epoch_{E}_batch_{i}.pt`` holding {'epoch', 'model_state_dict', 'optimizer_state_dict'}
(train.py:201-291).  What differs is the loop BODY: ``zero_grad -> model -> loss -> backward -> step``
(train.py:240-284) is one fused launch sequence (engine.HipTrainer): HIP kernels, gradients in one flat
buffer, RCCL all-reduce in buckets overlapped with the backward, one optimizer launch, hipGraph replay, no
per-step host sync (the reference syncs 8x per step through `.item()` and uploads to wandb every step).

Reference defects fixed on the way: undefined DEV / mp / time (train.py:136,152,212), no
--device flag, checkpoint keys saved with DDP's `module.` prefix, checkpoint dir join that drops the model
type, `git_hash` storing the function object; wandb is optional.
"""
import argparse
import logging
import os
import time
from datetime import timedelta
from typing import Dict, List

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler as DS

from ..engine import HipTrainer
from ..loss.DiffusionLossEvaluator import DiffusionLossEvaluator
from ..loss.RegressionLossEvaluator import RegressionLossEvaluator
from ._common import (add_additive_flags, add_component_flags, dtype_of, is_diffusion, open_dataset, pick_device)
from .abstract_command import MODEL_TYPES, AbstractCommand
from .utilities import get_git_hash, has_uncommitted_changes

DEV = 'dev'   # the dev split directory (analyze.py:80); the reference train.py uses an undefined name here


class TrainCommand(AbstractCommand):
    last_run_stats = None        # {"epoch", "steps", "seconds", "windows_per_s"} of the last finished training epoch

    def __init__(self):
        super().__init__()

    def register_subcommand(self, subparsers: argparse._SubParsersAction):
        p = subparsers.add_parser('train', help='Train a model on the AddBiomechanics dataset')
        p.add_argument('--dataset-home', type=str, default='../data', help='The path to the AddBiomechanics dataset.')
        p.add_argument('--no-wandb', action='store_true', default=False, help='Do not log this run to Weights and Biases.')
        p.add_argument('--model-type', type=str, default='feedforward', choices=MODEL_TYPES, help='The model to train.')
        p.add_argument('--output-data-format', type=str, default='all_frames', choices=['all_frames', 'last_frame'],
                       help='Output for all frames in a window or only the last frame.')
        p.add_argument('--checkpoint-dir', type=str, default='../checkpoints',
                       help='Where checkpoints are saved; training resumes from the latest one in this directory.')
        p.add_argument('--geometry-folder', type=str, default=None, help='Path to the Geometry folder with bone mesh data.')
        p.add_argument('--history-len', type=int, default=50, help='The number of timesteps of context in the inputs.')
        p.add_argument('--stride', type=int, default=5, help='The timestep gap between frames in the context window.')
        p.add_argument('--learning-rate', type=float, default=1e-4, help='The learning rate for weight updates.')
        p.add_argument('--dropout', action='store_true', help='Apply dropout?')
        p.add_argument('--dropout-prob', type=float, default=0.5, help='Dropout prob')
        p.add_argument('--hidden-dims', type=int, nargs='+', default=[512, 512], help='Hidden dims across layers.')
        p.add_argument('--batchnorm', action='store_true', help='Apply batchnorm?')
        p.add_argument('--activation', type=str, default='sigmoid', help='Which activation func?')
        p.add_argument('--epochs', type=int, default=10, help='The number of epochs to run training for.')
        p.add_argument('--opt-type', type=str, default='rmsprop', help='The optimizer used to adapt the weights.')
        p.add_argument('--batch-size', type=int, default=64, help='The batch size (per process).')
        p.add_argument('--short', action='store_true', help='Use very short datasets to test quickly.')
        p.add_argument('--data-loading-workers', type=int, default=1, help='Worker processes that load data.')
        add_component_flags(p, train_defaults=True)
        p.add_argument('--trial-filter', type=str, nargs='+', default=[""], help='What kind of trials to train/test on.')
        p.add_argument('--compute-report', action='store_true', default=False,
                       help='Compute inverse dynamics reports during loss evaluation.')
        add_additive_flags(p)
        p.add_argument('--max-steps', type=int, default=0, help='Stop each epoch after this many batches (0 = all).')
        p.add_argument('--report-every', type=int, default=1000, help='Batches between reports / checkpoints.')
        p.add_argument('--eager', action='store_true',
                       help='Reference-style loop (autograd node + torch.optim + torch DDP) instead of the fused trainer.')
        p.add_argument('--no-graph', action='store_true', help='Do not replay the step from hipGraphs.')
        p.add_argument('--bucket-mb', type=float, default=13.0, help='Gradient all-reduce bucket size (MiB).')
        p.add_argument('--window-cache', type=str, default=None,
                       help='Packed-window file (data/WindowCache.py): loaded if it exists, else built from the training '
                            'set and saved; training then runs from an on-device window cache (one gather launch per '
                            'batch instead of the DataLoader pipeline).  Regression models: packed-window file '
                            '(data/WindowCache.py); diffusion models: a .npy of [N, T, D] motion windows, and the '
                            "step's timesteps / noise are drawn on the device by the same launch; the word `hbm` with "
                            '--synthetic-windows draws the synthetic windows straight into HBM (no file).')
        p.add_argument('--max-dev-steps', type=int, default=0,
                       help='Evaluate at most this many dev batches before each epoch (0 = follow --max-steps).')
        p.add_argument('--loss-every', type=int, default=1,
                       help='[diffusion, --window-cache] keep the device-resident loss of every Nth step for the report.')
        p.add_argument('--seed', type=int, default=None,
                       help='torch.manual_seed for this run (initial weights, dropout masks, diffusion noise).')

    # ------------------------------------------------------------------------------------------
    def run(self, args: argparse.Namespace):
        if 'command' in args and args.command != 'train':
            return False
        model_type: str = args.model_type
        checkpoint_dir: str = os.path.join(os.path.abspath(args.checkpoint_dir), model_type)   # as analyze.py:57
        history_len, stride = args.history_len, args.stride
        log_to_wandb: bool = not args.no_wandb
        diffusion = is_diffusion(model_type)

        if getattr(args, 'seed', None) is not None:
            torch.manual_seed(args.seed)
        geometry = self.ensure_geometry(args.geometry_folder)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL (before the first HIP call)
        from .. import ddp_probe
        ddp_probe.prepare_env()                                      # preconditions of captured collectives (ddp_probe.py)
        device = pick_device(args)
        world_size = int(os.environ.get('WORLD_SIZE', '1'))
        distributed = world_size > 1
        if distributed:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if os.environ.get('IB_BENCH_REHEARSAL') == '1':
                dist.init_process_group(backend="gloo", timeout=timedelta(hours=1))
            else:
                dist.init_process_group(backend="nccl", timeout=timedelta(hours=1), device_id=device)   # RCCL (train.py:99)
        rank = dist.get_rank() if distributed else 0
        print(f"Running on {world_size} GPUs.")
        print(f"Current device being used for model training and loss evaluation: {device}.")
        if has_uncommitted_changes():
            logging.error("UNCOMMITTED CHANGES IN REPO! This will make it hard to replicate this experiment later")

        wandb = None
        if log_to_wandb:
            try:
                import wandb as _wandb
                wandb = _wandb
                config = dict(args.__dict__)
                config["git_hash"] = get_git_hash()
                group = os.getenv('WANDB_RUN_GROUP', f'ddp_{wandb.util.generate_id()}')
                wandb.init(project="addbiomechanics-baseline", config=config, group=group)
            except ImportError:
                logging.warning("wandb is not installed: continuing without uploads (same as --no-wandb)")
                log_to_wandb = False

        print("Initializing training set...")
        train_dataset = open_dataset(args, 'train', history_len, stride, args.output_data_format, geometry)
        print("Initializing dev set...")
        dev_dataset = open_dataset(args, DEV, history_len, stride, args.output_data_format, geometry)
        # DistributedSampler(shuffle=False, drop_last=True): rank r takes indices r, r+world, ... (train.py:143,149)
        mk = lambda ds: DataLoader(ds, batch_size=args.batch_size, shuffle=False, num_workers=args.data_loading_workers,
                                   persistent_workers=args.data_loading_workers > 0, pin_memory=True, drop_last=diffusion,
                                   sampler=DS(ds, num_replicas=world_size, rank=rank, shuffle=False, drop_last=True))
        train_dataloader, dev_dataloader = mk(train_dataset), mk(dev_dataset)

        print("Initializing model...")
        window = history_len // stride if stride > 1 else history_len
        model = self.get_model(getattr(train_dataset, 'num_dofs', 23), getattr(train_dataset, 'num_contact_bodies', 2),
                               model_type, history_len=history_len, stride=stride, hidden_dims=args.hidden_dims,
                               activation=args.activation, batchnorm=args.batchnorm, dropout=args.dropout,
                               dropout_prob=args.dropout_prob, root_history_len=10,
                               output_data_format=args.output_data_format, device=device, compute_dtype=dtype_of(args),
                               feat_dim=getattr(train_dataset, 'feat', args.feat_dim), window=window).to(device)
        if not any(p.requires_grad for p in model.parameters()):
            print("No parameters to optimize. Skipping training loop.")
            return False

        if diffusion:
            train_eval, dev_eval = DiffusionLossEvaluator('train'), DiffusionLossEvaluator(DEV)
        else:
            train_eval = RegressionLossEvaluator(dataset=train_dataset, split='train', device=device)
            dev_eval = RegressionLossEvaluator(dataset=dev_dataset, split=DEV, device=device)

        trainer, optimizer, ddp_model = None, None, model
        if args.eager:
            if distributed:
                from torch.nn.parallel import DistributedDataParallel as DDP
                model.ensure_packed()
                ddp_model = DDP(model, device_ids=[device.index], output_device=device.index)
            optimizer = make_torch_optimizer(args.opt_type, model.parameters(), args.learning_rate)
        else:
            trainer = HipTrainer(model, "diffusion" if diffusion else "regression", args.opt_type, args.learning_rate,
                                 args=args, use_graph=not args.no_graph, bucket_mb=args.bucket_mb)

        if getattr(args, 'loss_every', 1) < 1:
            raise SystemExit("--loss-every must be >= 1")
        if args.window_cache == 'hbm' and not (diffusion and getattr(args, 'synthetic_windows', 0) > 0):
            raise SystemExit("--window-cache hbm names the device-resident synthetic table: it needs a diffusion model type "
                             "and --synthetic-windows N (a cache FILE takes a path)")
        cache = None
        if args.window_cache and trainer is not None:
            from ..data.WindowCache import DeviceMotionCache, DeviceWindowCache, PackedWindows, wait_for_file
            # rank 0 packs and writes (atomically: temporary file + rename); the other ranks POLL for the finished file and
            # only then enter the barrier -- packing a full training set can outlast the process group's collective timeout
            # (~10 min under RCCL), so no collective may span it; none of them can open a file that is still being written
            in_hbm = diffusion and args.window_cache == 'hbm' and getattr(args, 'synthetic_windows', 0) > 0
            # the failure marker carries this launch's id (every rank of a torchrun launch sees the same one): the marker a
            # previous, failed launch left behind has another name and is never mistaken for this run's
            run_id = os.environ.get("TORCHELASTIC_RUN_ID", "none") + "." + os.environ.get("MASTER_PORT", "0")
            failed_marker = f"{args.window_cache}.failed.{run_id}"
            if in_hbm:
                pass                  # synthetic windows are drawn straight into HBM below: no file, no host pass
            elif not os.path.exists(args.window_cache) and rank == 0:
                print(f"Packing {len(train_dataset)} training windows into {args.window_cache} ...")
                try:
                    if diffusion:
                        save_motion_windows(train_dataset, args.window_cache)
                    else:
                        PackedWindows.from_dataset(train_dataset, workers=args.data_loading_workers).save(args.window_cache)
                except BaseException as exc:      # the waiting ranks poll for this instead of sitting out their timeout
                    with open(failed_marker, "w") as f:
                        f.write(repr(exc))
                    raise
            if distributed and not in_hbm:
                wait_for_file(args.window_cache, float(os.environ.get("IB_WINDOW_CACHE_WAIT_S", 6 * 3600)),
                              failed=failed_marker)
                dist.barrier()
            if in_hbm:
                cache = DeviceMotionCache.synthetic(len(train_dataset), window, train_dataset.feat, device,
                                                    model.compute_dtype, seed=0)
            elif diffusion:
                import numpy as np
                cache = DeviceMotionCache(torch.from_numpy(np.load(args.window_cache, mmap_mode='r')), device,
                                          model.compute_dtype)
            else:
                cache = DeviceWindowCache(PackedWindows.load(args.window_cache), device)
            print(f"[rank={rank}] window cache: {len(cache)} windows, "
                  f"{cache.table.numel() * cache.table.element_size() / 2**20:.1f} MiB in HBM")

        epoch_checkpoint, _ = self.load_latest_checkpoint(model, optimizer=trainer if trainer is not None else optimizer,
                                                          checkpoint_dir=checkpoint_dir)
        from .. import hip
        noise_seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF

        def device_batch(x0, stream_id, draw):
            """x0 [B, T, D] (host or device) -> (x0, t, eps) on the device in the model's dtype; t / eps from the
            counter-based generator (csrc/noise.hip), counter `draw`: the dev-set evaluation and the --eager loop draw like
            the fused trainer does, nothing random is made on the host.  Two counters: the dev set's starts at 0 before
            every epoch (the same noise every time: comparable reports); the --eager training loop's is the global step
            epoch * batches-per-epoch + i, which only ever increases and is right again after a resume -- a window meets
            fresh (t, eps) in every epoch, as with the host generator of the reference-style loop"""
            xd = x0.to(device, model.compute_dtype).contiguous()
            ed = torch.empty_like(xd)
            td = torch.empty(xd.shape[0], dtype=torch.int64, device=device)
            hip.diffusion_draw(noise_seed, step=int(draw), stream_id=stream_id, eps=ed, t=td,
                               num_train_steps=model.num_train_steps)
            return xd, td, ed

        adopted = False
        prev_stream = None
        for epoch in range(epoch_checkpoint + 1, args.epochs):
            dev_dataloader.sampler.set_epoch(epoch)
            train_dataloader.sampler.set_epoch(epoch)
            print(f'[rank={rank}] Evaluating Dev Set Before Epoch {epoch}')
            with torch.no_grad():
                model.eval()
                for i, batch in enumerate(dev_dataloader):
                    if diffusion:
                        xd, td, ed = device_batch(batch, 0x80000000 | rank, i)   # the same noise before every epoch
                        tabs = model.tables(device)
                        xt = torch.empty_like(xd)
                        hip.q_sample(xd, ed, td, tabs.sqrt_ab, tabs.sqrt_1mab, xt)
                        dev_eval(model(xt, td), ed)
                    else:
                        inputs, labels, subj, trial = batch
                        dev_eval(inputs, model(inputs), labels, subj, trial, args, compute_report=args.compute_report)
                    if (args.max_dev_steps or args.max_steps) and i + 1 >= (args.max_dev_steps or args.max_steps):
                        break
                print(f'[rank={rank}] Dev Set Evaluation: ')
                dev_eval.print_report(args, log_to_wandb=log_to_wandb) if not diffusion else dev_eval.print_report()
            if distributed:
                dist.barrier()
            print(f'[rank={rank}] Running Training Epoch {epoch}')
            model.train()
            if trainer is not None and not adopted:
                prev_stream = trainer.adopt_stream()   # the loop's device work runs on the trainer's stream: no per-step hand-over
                adopted = True
            if cache is not None:
                train_batches = list(cache.batches(args.batch_size, rank=rank, world=world_size))
            else:
                train_batches = train_dataloader
            n_batches = len(train_batches)
            t_epoch, steps_epoch = time.perf_counter(), 0
            for i, batch in enumerate(train_batches):
                steps_epoch += 1
                if cache is not None and diffusion:
                    loss = trainer.step_drawn(cache, batch)           # gather + draw + fused step: all on the device
                    if (i + 1) % args.loss_every == 0:
                        train_eval.losses.append(loss.detach().clone())
                elif cache is not None:
                    trainer.step_windows(cache, batch)
                    train_eval.record_result(trainer.result)
                elif diffusion:
                    if trainer is not None:
                        loss = trainer.step_x0(batch.to(device, non_blocking=True))
                        train_eval.losses.append(loss.detach().clone())
                    else:
                        optimizer.zero_grad()
                        tabs = model.tables(device)
                        xd, td, ed = device_batch(batch, rank, epoch * n_batches + i)
                        xt = torch.empty_like(xd)
                        hip.q_sample(xd, ed, td, tabs.sqrt_ab, tabs.sqrt_1mab, xt)
                        loss = train_eval(ddp_model(xt, td), ed)
                        loss.backward()
                        optimizer.step()
                else:
                    inputs, labels, subj, trial = batch
                    if trainer is not None:
                        trainer.step((inputs, labels))
                        train_eval.record_result(trainer.result)
                    else:
                        optimizer.zero_grad()
                        loss = train_eval(inputs, ddp_model(inputs), labels, subj, trial, args,
                                          compute_report=args.compute_report and (i % 100 == 0))
                        loss.backward()
                        optimizer.step()
                last = (i == n_batches - 1) or (args.max_steps and i + 1 >= args.max_steps)
                if (i + 1) % 100 == 0 or last:
                    logging.info(f'  - [rank={rank}] Batch {i + 1}/{n_batches}')
                if (i + 1) % args.report_every == 0 or last:
                    logging.info(f'[rank={rank}] Batch {i} Training Set Evaluation: ')
                    train_eval.print_report(args, reset=False) if not diffusion else train_eval.print_report(reset=False)
                    if rank == 0:
                        save_checkpoint(checkpoint_dir, epoch, i, model, trainer if trainer is not None else optimizer)
                if last:
                    break
            if device.type == 'cuda':
                torch.cuda.synchronize(device)
            dt_epoch = time.perf_counter() - t_epoch
            TrainCommand.last_run_stats = {"epoch": epoch, "steps": steps_epoch, "seconds": dt_epoch,
                                           "windows_per_s": steps_epoch * args.batch_size * world_size / max(dt_epoch, 1e-9)}
            print(f"[rank={rank}] epoch {epoch}: {steps_epoch} steps in {dt_epoch:.3f} s = "
                  f"{self.last_run_stats['windows_per_s']:.0f} windows/s (all ranks, reports and checkpoints included)")
            logging.info('-' * 80)
            logging.info(f'[rank={rank}] Epoch {epoch}/{args.epochs} Training Set Evaluation: ')
            train_eval.print_report(args, log_to_wandb=log_to_wandb) if not diffusion else train_eval.print_report()
            logging.info('-' * 80)

        if prev_stream is not None:                 # in-process callers (tests, tools) get their stream back
            torch.cuda.current_stream().synchronize()
            torch.cuda.set_stream(prev_stream)
        if wandb is not None:
            wandb.finish()
        if distributed:
            dist.destroy_process_group()
        return True


def save_motion_windows(dataset, path: str, chunk: int = 4096):
    """the [N, T, D] motion windows of a data set as one float32 .npy (written under a temporary name, then renamed)"""
    import numpy as np
    n = len(dataset)
    first = dataset[0]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = f"{path}.tmp.{os.getpid()}.npy"
    out = np.lib.format.open_memmap(tmp, mode='w+', dtype=np.float32, shape=(n,) + tuple(first.shape))
    for i in range(n):
        out[i] = dataset[i].to(torch.float32).numpy()
    out.flush()
    del out
    os.replace(tmp, path)


def make_torch_optimizer(opt_type: str, params, lr: float):
    table = {'adagrad': torch.optim.Adagrad, 'adam': torch.optim.Adam, 'sgd': torch.optim.SGD,
             'rmsprop': torch.optim.RMSprop, 'adadelta': torch.optim.Adadelta, 'adamax': torch.optim.Adamax}
    if opt_type not in table:
        logging.error('Invalid optimizer type: ' + opt_type)
        assert (False)
    return table[opt_type](params, lr=lr)


def save_checkpoint(checkpoint_dir: str, epoch: int, batch: int, model, opt):
    """file grammar of train.py:271-278; keys are saved WITHOUT DDP's `module.` prefix"""
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = f"{checkpoint_dir}/epoch_{epoch}_batch_{batch}.pt"
    osd = opt.optimizer_state_dict() if hasattr(opt, 'optimizer_state_dict') else opt.state_dict()
    torch.save({'epoch': epoch,
                'model_state_dict': {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                'optimizer_state_dict': osd}, path)
    return path
