"""Packed motion windows + on-device window cache.

The reference builds every training window on the host through nimblephysics: ``AddBiomechanicsDataset.__getitem__``
(src/data/AddBiomechanicsDataset.py:161-285) returns ``(inputs: Dict[10 keys -> [F, c]], labels: Dict[.. -> [F, c']],
subject_index, trial)``, the DataLoader collates ~17 small tensors per window, and the model concatenates the 10 input
keys again (src/models/FeedForwardRegressionBaseline.py:97-108).  At GPU step rates that pipeline is the bottleneck.
Its own escape hatch is ``pickle-data`` (src/cli/pickle_data.py:52-64): ``torch.save`` of a LIST of those tuples per
block, reloaded by ``PickledDataset`` (src/data/PickledDataset.py:16-25).

Here a window is ONE contiguous fp32 row::

    [ model input, frame-major  F x sum(widths), zero-padded to a multiple of 4 values |
      labels, key-major: cop F'x6 | force F'x6 | torque F'x6 | wrench F'x12 ]

* ``PackedWindows``  host side: built from any source of reference-layout tuples (a reference dataset object, a
  ``torch.load``-ed pickle block, ``SyntheticWindowDataset``); stored as one raw little-endian file with a JSON header
  (``save`` / ``load``; ``load`` memory-maps, so files larger than RAM stream from disk).
* ``DeviceWindowCache``  the rows in HBM (uploaded through a pinned staging buffer, chunk by chunk) + rank-sharded
  batch index generation with the reference's sampler semantics (``DistributedSampler(shuffle=False, drop_last=True)``,
  src/cli/train.py:143-150).  ``HipTrainer.step_windows(cache, idx)`` gathers a batch with ONE launch
  (``ib_gather_windows``) straight into the model's input tensor and the loss kernel's label tensors.

Label values are stored as given (the reference loader has already divided forces / torques / wrenches by the subject's
mass, AddBiomechanicsDataset.py:248-261)."""
import json
import os
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .AddBiomechanicsDataset import INPUT_KEY_ORDER, LOSS_KEY_ORDER, LOSS_KEY_WIDTHS, packed_row_layout

MAGIC = b"IBWINDOWS1\n"
HEADER_BYTES = 4096


def wait_for_file(path: str, timeout_s: float, poll_s: float = 2.0, failed: str = None) -> None:
    """Sleep-poll until `path` exists.  PackedWindows.save renames the finished file into place, so existence means
    complete.  Used by the ranks that do not pack (cli/train.py): waiting here instead of inside a collective keeps the
    process group's watchdog timeout out of a phase whose length depends on the data set.  `failed`: the marker file the
    packing rank of THIS launch writes on an exception (its name carries the launch id: a marker left behind by an earlier,
    failed launch is a different file and is ignored)."""
    import time
    failed = failed if failed is not None else path + ".failed"
    t0 = time.monotonic()
    while not os.path.exists(path):
        if os.path.exists(failed):                     # the packing rank died (cli/train.py writes it on any exception)
            raise RuntimeError(f"window cache {path!r}: the packing rank failed: " + open(failed).read()[:500])
        if time.monotonic() - t0 > timeout_s:
            raise TimeoutError(f"window cache {path!r} did not appear within {timeout_s:.0f} s (is the packing rank alive?)")
        time.sleep(poll_s)


class _RowView(torch.utils.data.Dataset):
    """index -> (packed row, subject, trial): lets DataLoader workers run `window_row`"""

    def __init__(self, dataset, n: int, row_elems: int, widths):
        self.dataset, self.n, self.row_elems, self.widths = dataset, n, row_elems, list(widths)

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, i: int):
        row = np.empty(self.row_elems, dtype=np.float32)
        s, t = self.dataset.window_row(i, row, self.widths)
        return torch.from_numpy(row), s, t


class PackedWindows:
    """N windows as one fp32 matrix ``rows[N, x_elems + y_elems]`` (+ subject / trial indices)."""

    def __init__(self, rows: np.ndarray, frames: int, out_frames: int, input_widths: Sequence[int],
                 subjects: Optional[np.ndarray] = None, trials: Optional[np.ndarray] = None):
        self.rows, self.frames, self.out_frames = rows, int(frames), int(out_frames)
        self.input_widths = [int(w) for w in input_widths]
        # 1470 input values at the reference defaults -> 1472; 'last_frame' labels 6 | 6 | 6 | 12 -> 8 | 8 | 8 | 12
        self.x_elems, self.x_pad, self.label_elems, self.label_pad, row_elems = \
            packed_row_layout(self.frames, self.out_frames, self.input_widths)
        if rows.ndim != 2 or rows.dtype != np.float32 or rows.shape[1] != row_elems:
            raise ValueError("PackedWindows: rows must be float32 [N, x_pad + padded label elems]")
        n = rows.shape[0]
        self.subjects = np.zeros(n, np.int32) if subjects is None else np.asarray(subjects, np.int32)
        self.trials = np.zeros(n, np.int32) if trials is None else np.asarray(trials, np.int32)

    def __len__(self) -> int:
        return self.rows.shape[0]

    # ---- building -------------------------------------------------------------------------------------------
    @staticmethod
    def row_of(inputs: Dict[str, torch.Tensor], labels: Dict[str, torch.Tensor]) -> np.ndarray:
        """one window -> its packed row: inputs concatenated per frame in the model's key order
        (FeedForwardRegressionBaseline.py:97-107), labels key-major in the loss order"""
        pad4 = lambda v: torch.cat([v, torch.zeros((-v.numel()) % 4, dtype=torch.float32)])
        x = torch.cat([inputs[k].to(torch.float32).reshape(inputs[k].shape[0], -1) for k in INPUT_KEY_ORDER], dim=-1)
        y = [pad4(labels[k].to(torch.float32).reshape(-1)) for k in LOSS_KEY_ORDER]
        return torch.cat([pad4(x.reshape(-1))] + y).numpy()

    @classmethod
    def from_windows(cls, windows: Iterable[Tuple], limit: Optional[int] = None) -> "PackedWindows":
        """`windows`: anything that yields / indexes the reference tuple layout (a dataset, a loaded pickle block)"""
        it = (windows[i] for i in range(len(windows))) if hasattr(windows, "__getitem__") and hasattr(windows, "__len__") \
            else iter(windows)
        rows, subj, trial = [], [], []
        geo = None
        for item in it:
            inputs, labels = item[0], item[1]
            g = (inputs[INPUT_KEY_ORDER[0]].shape[0], labels[LOSS_KEY_ORDER[0]].shape[0],
                 tuple(int(np.prod(inputs[k].shape[1:])) for k in INPUT_KEY_ORDER))
            if geo is None:
                geo = g
            elif g != geo:
                raise ValueError(f"PackedWindows: window geometry changed ({g} vs {geo})")
            rows.append(cls.row_of(inputs, labels))
            subj.append(int(item[2]) if len(item) > 2 else 0)
            trial.append(int(item[3]) if len(item) > 3 else 0)
            if limit is not None and len(rows) >= limit:
                break
        if not rows:
            raise ValueError("PackedWindows: no windows")
        return cls(np.stack(rows).astype(np.float32, copy=False), geo[0], geo[1], geo[2], subj, trial)

    @classmethod
    def from_dataset(cls, dataset, limit: Optional[int] = None, workers: int = 0, chunk: int = 256) -> "PackedWindows":
        """Pack a window data set.  A loader that can write packed rows itself (`AddBiomechanicsDataset.window_row`)
        fills the row matrix in place -- no per-window tensor dicts, no collate; `workers` > 0 spreads the reads over
        DataLoader worker processes (each re-opens its `.b3d` handles).  Anything else goes through `from_windows`."""
        if not hasattr(dataset, "window_row"):
            return cls.from_windows(dataset, limit)
        n = len(dataset) if limit is None else min(int(limit), len(dataset))
        frames, out_frames, widths = dataset.row_geometry()
        rows = np.empty((n, packed_row_layout(frames, out_frames, widths)[4]), dtype=np.float32)
        subj, trial = np.zeros(n, np.int32), np.zeros(n, np.int32)
        if workers <= 0:
            for i in range(n):
                subj[i], trial[i] = dataset.window_row(i, rows[i], widths)
        else:
            view = _RowView(dataset, n, rows.shape[1], widths)
            loader = torch.utils.data.DataLoader(view, batch_size=chunk, shuffle=False, num_workers=workers)
            a = 0
            for r, s, t in loader:
                b = a + r.shape[0]
                rows[a:b], subj[a:b], trial[a:b] = r.numpy(), s.numpy(), t.numpy()
                a = b
        return cls(rows, frames, out_frames, widths, subj, trial)

    @classmethod
    def from_pickled_blocks(cls, paths: Sequence[str]) -> "PackedWindows":
        """the reference's `pickle-data` blocks: each file is torch.save(list of window tuples)"""
        packs = [cls.from_windows(torch.load(p, map_location="cpu", weights_only=False)) for p in paths]
        first = packs[0]
        for p in packs[1:]:
            if (p.frames, p.out_frames, p.input_widths) != (first.frames, first.out_frames, first.input_widths):
                raise ValueError("PackedWindows: blocks with different window geometry")
        return cls(np.concatenate([p.rows for p in packs]), first.frames, first.out_frames, first.input_widths,
                   np.concatenate([p.subjects for p in packs]), np.concatenate([p.trials for p in packs]))

    # ---- the reference tuple back from a row (analyze / tests) ------------------------------------------------
    def window(self, i: int) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor], int, int]:
        r = torch.from_numpy(np.array(self.rows[i]))
        x = r[:self.x_elems].reshape(self.frames, sum(self.input_widths))
        inputs, off = {}, 0
        for k, w in zip(INPUT_KEY_ORDER, self.input_widths):
            inputs[k] = x[:, off:off + w].clone()
            off += w
        labels, off = {}, self.x_pad
        for k, c, e, ep in zip(LOSS_KEY_ORDER, LOSS_KEY_WIDTHS, self.label_elems, self.label_pad):
            labels[k] = r[off:off + e].reshape(self.out_frames, c).clone()
            off += ep
        return inputs, labels, int(self.subjects[i]), int(self.trials[i])

    # ---- file format ------------------------------------------------------------------------------------------
    def save(self, path: str):
        hdr = {"version": 1, "windows": len(self), "row_elems": int(self.rows.shape[1]), "frames": self.frames,
               "out_frames": self.out_frames, "input_widths": self.input_widths, "input_keys": INPUT_KEY_ORDER,
               "label_keys": LOSS_KEY_ORDER, "label_widths": LOSS_KEY_WIDTHS, "dtype": "<f4"}
        blob = json.dumps(hdr).encode()
        if len(MAGIC) + len(blob) > HEADER_BYTES:
            raise ValueError("PackedWindows: header too large")
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        # written under a temporary name and renamed: a reader (another rank, a later run) never sees a partial file
        tmp = f"{path}.tmp.{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write((MAGIC + blob).ljust(HEADER_BYTES, b" "))
            np.ascontiguousarray(self.rows, dtype="<f4").tofile(f)
            np.ascontiguousarray(self.subjects, dtype="<i4").tofile(f)
            np.ascontiguousarray(self.trials, dtype="<i4").tofile(f)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)

    @classmethod
    def load(cls, path: str, mmap: bool = True) -> "PackedWindows":
        with open(path, "rb") as f:
            head = f.read(HEADER_BYTES)
        if not head.startswith(MAGIC):
            raise ValueError(f"{path}: not a packed-window file")
        hdr = json.loads(head[len(MAGIC):].decode().strip())
        if hdr.get("version") != 1 or hdr.get("dtype") != "<f4" or hdr.get("input_keys") != INPUT_KEY_ORDER \
                or hdr.get("label_keys") != LOSS_KEY_ORDER:
            raise ValueError(f"{path}: unsupported packed-window header {hdr}")
        n, re = hdr["windows"], hdr["row_elems"]
        size = os.path.getsize(path)
        if size != HEADER_BYTES + 4 * n * re + 8 * n:
            raise ValueError(f"{path}: truncated (size {size})")
        if mmap:
            rows = np.memmap(path, dtype="<f4", mode="r", offset=HEADER_BYTES, shape=(n, re))
        else:
            rows = np.fromfile(path, dtype="<f4", count=n * re, offset=HEADER_BYTES).reshape(n, re)
        tail = np.fromfile(path, dtype="<i4", count=2 * n, offset=HEADER_BYTES + 4 * n * re)
        return cls(rows, hdr["frames"], hdr["out_frames"], hdr["input_widths"], tail[:n], tail[n:])


def sharded_batches(n: int, device, batch_size: int, rank: int = 0, world: int = 1, drop_last: bool = True,
                    shuffle_seed: Optional[int] = None) -> Iterator[torch.Tensor]:
    """int64 index tensors ON THE DEVICE.  Sharding = DistributedSampler(shuffle=False, drop_last=True) as the reference
    constructs it (train.py:143,149): rank r owns windows r, r + world, ...; `shuffle_seed` permutes the rank's share per
    epoch (the reference never shuffles)."""
    per = n // world if drop_last else (n + world - 1) // world
    own = torch.arange(rank, rank + per * world, world, dtype=torch.int64) % n
    if shuffle_seed is not None:
        own = own[torch.randperm(per, generator=torch.Generator().manual_seed(shuffle_seed))]
    own = own.to(device)
    stop = per - per % batch_size if drop_last else per
    for a in range(0, stop, batch_size):
        yield own[a:a + batch_size]


class DeviceMotionCache:
    """[BUILD-DEFINED] motion windows ``[N, T, D]`` for the diffusion denoisers, resident in HBM in the model's compute
    dtype (bf16: 30 KB per 50 x 300 window, 9.6 M windows per 288 GB); rows pitched to 8 values so every window starts
    on a 16-byte boundary.  `HipTrainer.step_drawn(cache, idx)` gathers a batch out of it and draws that step's timesteps
    and noise in the same launch (`ib_diffusion_draw`): the training loop touches no host tensor."""

    @classmethod
    def synthetic(cls, num_windows: int, window: int, feat: int, device, dtype: torch.dtype = torch.bfloat16,
                  seed: int = 0) -> "DeviceMotionCache":
        """x0 ~ N(0,1) synthetic windows drawn straight into HBM by the counter-based generator
        (csrc/noise.hip) -- a million 50 x 300 windows (30 GB of bf16) in milliseconds, nothing staged through the host"""
        from .. import hip
        self = cls.__new__(cls)
        self.device = torch.device(device)
        self.window, self.feat = int(window), int(feat)
        per = self.window * self.feat
        self.pitch = (per + 7) // 8 * 8
        self.table = torch.zeros((int(num_windows), self.pitch), dtype=dtype, device=self.device)
        rows = max(1, (1 << 31) // self.pitch)               # a draw call addresses < 2^32 four-value blocks
        for a in range(0, int(num_windows), rows):
            part = self.table[a:a + rows]
            hip.diffusion_draw(seed, step=a // rows, stream_id=0x20000000, eps=part)
        if self.pitch != per:
            self.table[:, per:].zero_()
        return self

    def __init__(self, windows, device, dtype: torch.dtype = torch.bfloat16, chunk_windows: int = 2048):
        self.device = torch.device(device)
        n = len(windows)
        if n == 0:
            raise ValueError("DeviceMotionCache: no windows")
        first = windows[0]
        if first.dim() != 2:
            raise ValueError("DeviceMotionCache: every window must be a [T, D] matrix")
        self.window, self.feat = int(first.shape[0]), int(first.shape[1])
        per = self.window * self.feat
        self.pitch = (per + 7) // 8 * 8
        self.table = torch.zeros((n, self.pitch), dtype=dtype, device=self.device)
        pin = self.device.type == "cuda"
        stage = torch.zeros((min(chunk_windows, n), per), dtype=torch.float32, pin_memory=pin)
        for a in range(0, n, stage.shape[0]):
            b = min(n, a + stage.shape[0])
            if isinstance(windows, torch.Tensor):
                stage[:b - a].copy_(windows[a:b].reshape(b - a, per))
            else:
                for i in range(a, b):
                    w = windows[i]
                    if tuple(w.shape) != (self.window, self.feat):
                        raise ValueError(f"DeviceMotionCache: window {i} is {tuple(w.shape)}, expected "
                                         f"{(self.window, self.feat)}")
                    stage[i - a].copy_(w.reshape(per))
            self.table[a:b, :per].copy_(stage[:b - a])      # fp32 -> compute dtype on the device (one rounding)

    def __len__(self) -> int:
        return self.table.shape[0]

    def batches(self, batch_size: int, rank: int = 0, world: int = 1, drop_last: bool = True,
                shuffle_seed: Optional[int] = None) -> Iterator[torch.Tensor]:
        return sharded_batches(len(self), self.device, batch_size, rank, world, drop_last, shuffle_seed)


class DeviceWindowCache:
    """The packed rows resident in HBM (fp32: 12.7 KB per window at the reference defaults, i.e. 22 M windows per
    288 GB) + per-rank batch index generation.  Upload goes through one pinned staging buffer, `chunk_windows` rows at
    a time, so a memory-mapped pack larger than host RAM can still be cached."""

    def __init__(self, pack: PackedWindows, device, chunk_windows: int = 65536):
        self.pack, self.device = pack, torch.device(device)
        self.frames, self.out_frames = pack.frames, pack.out_frames
        self.x_elems, self.label_elems = pack.x_elems, list(pack.label_elems)
        n, re = pack.rows.shape
        self.table = torch.empty((n, re), dtype=torch.float32, device=self.device)
        pin = self.device.type == "cuda"
        stage = torch.empty((min(chunk_windows, n), re), dtype=torch.float32, pin_memory=pin)
        for a in range(0, n, stage.shape[0]):
            b = min(n, a + stage.shape[0])
            stage[:b - a].copy_(torch.from_numpy(np.ascontiguousarray(pack.rows[a:b])))
            self.table[a:b].copy_(stage[:b - a], non_blocking=False)

    def __len__(self) -> int:
        return self.table.shape[0]

    def batches(self, batch_size: int, rank: int = 0, world: int = 1, drop_last: bool = True,
                shuffle_seed: Optional[int] = None) -> Iterator[torch.Tensor]:
        """int64 index tensors ON THE DEVICE.  Sharding = DistributedSampler(shuffle=False, drop_last=True) as the
        reference constructs it (train.py:143,149): rank r owns windows r, r + world, ...; `shuffle_seed` permutes the
        rank's share per epoch (the reference never shuffles)."""
        return sharded_batches(len(self), self.device, batch_size, rank, world, drop_last, shuffle_seed)

    def label_shapes(self, B: int) -> List[Tuple[int, int, int]]:
        return [(B, self.out_frames, c) for c in LOSS_KEY_WIDTHS]
