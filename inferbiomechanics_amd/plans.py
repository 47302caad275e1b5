"""Explicit forward / backward launch plans over the HIP kernels.

A *plan* is the MI355X-native replacement of "autograd over stock torch ops" for one model of the
hot path: a fixed sequence of C-ABI launches over buffers that stay resident in HBM (allocated once
per (plan, batch shape), so a warmed-up plan allocates nothing and can be captured into a hipGraph).
Activations needed by the backward are kept in those buffers; nothing is re-derived by a tracing
compiler.  Parameters are read through ``ParamSource`` (fp32 masters, or their bf16 shadows in
throughput mode); gradients are written straight into caller-provided fp32 tensors (the flat grad
buffer of the trainer, or scratch tensors handed to autograd).

Reference arithmetic implemented here (file:line relative to the reference root):
  * DenseStackPlan        -- FeedForwardBaseline.net, src/models/FeedForwardRegressionBaseline.py:65-77,113
  * TransformerLayerPlan  -- TransformerLayer.forward, src/models/TransformerBaseline.py:24-38
  * DenoiserMLPPlan / DenoiserTransformerPlan -- [BUILD-DEFINED] diffusion denoisers
"""
from __future__ import annotations

import os
import zlib
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import hip
from ._tuning import tuning as TU



def mask_seed(site: int) -> int:
    """Seed of a plan's dropout masks: the per-site constant mixed with torch's global seed (so torch.manual_seed(s)
    selects the mask stream as it does for nn.Dropout in the reference) and the distributed rank (data-parallel
    ranks must not drop the same elements).  Read when the plan is built; masks stay reproducible within a run because
    the backward regenerates them from the same (seed, step)."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    x = (int(site) * 0x9E3779B1 + (torch.initial_seed() & 0xFFFFFFFF) * 0x85EBCA6B + rank * 0xC2B2AE35) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    return x & 0x7FFFFFF0            # headroom for the "+ site index" the plans add


class Buffers:
    """Named HBM buffers, allocated on first use and reused afterwards (graph-capture safe)."""

    def __init__(self, device):
        self.device = device
        self._b: Dict[tuple, torch.Tensor] = {}

    def get(self, name: str, shape, dtype, zero: bool = False) -> torch.Tensor:
        """zero: cleared ONCE, when the buffer is created (pad columns that no launch ever writes)"""
        key = (name, tuple(int(s) for s in shape), dtype)
        t = self._b.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(key[1], dtype=dtype, device=self.device)
            self._b[key] = t
        return t

    def bytes(self, name: str, nbytes: int) -> torch.Tensor:
        n = max(int(nbytes), 16)
        key = (name, "bytes")
        t = self._b.get(key)
        if t is None or t.numel() < n:
            t = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._b[key] = t
        return t

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._b.values())


class ParamSource:
    """name -> tensor in the plan's compute dtype for matrices (``w``), fp32 for vectors (``v``);
    ``g`` -> the fp32 tensor that receives d loss / d param."""

    def __init__(self, w: Callable[[str], torch.Tensor], v: Callable[[str], torch.Tensor],
                 g: Optional[Callable[[str], torch.Tensor]] = None,
                 ready: Optional[Callable[[str], None]] = None, flush: Optional[Callable[[], None]] = None):
        self.w, self.v, self.g = w, v, g
        # called by a plan at points where NO side stream is forked (after its joins): the trainer then launches the
        # all-reduce of every gradient bucket completed so far (it cuts the captured graph there)
        self.flush = flush if flush is not None else (lambda: None)
        # called by a plan's backward right after the launches that complete d loss / d param[name] are
        # enqueued, in the order given by the plan's ready_order(): lets the trainer start the RCCL
        # all-reduce of a gradient bucket while the rest of the backward is still running
        self.ready = ready if ready is not None else (lambda name: None)


class Branch:
    """A side stream forked from / joined into the current stream with events.  Launches issued inside `run`
    execute concurrently with what the main stream does next (successive `run`s of one Branch are ordered among
    themselves); under hipGraph capture the same event edges become a parallel branch of the graph.  Disabled
    (runs inline) off-GPU and when `on` is False."""

    def __init__(self, device, enabled: bool = True, name: str = ""):
        import os
        self.name = name
        off = [n for n in TU.no_branch.split(",") if n]
        enabled = enabled and name not in off and "all" not in off
        self.on = enabled and torch.device(device).type == "cuda" and not hip._dry_run
        self.stream = hip.new_stream(device) if self.on else None      # never torch's pool (shared with c10d)
        self._forks: List = []
        self._join = torch.cuda.Event() if self.on else None
        self._n = 0

    _depth = 0      # class-wide: > 0 while some enabled branch's launches are being issued

    def run(self, fn):
        if not self.on:
            fn()
            return
        if Branch._depth > 0:
            # a fork nested inside a forked stream crashed hipStreamEndCapture (round 1, gpurun_out/dbg_graph.err):
            # branches must be SIBLINGS of the main stream.  Refused in eager mode too, so the first warm-up step
            # already reports it instead of a later capture segfaulting.
            raise hip.HipError(f"Branch '{self.name}': fork nested inside another forked stream (branches must be "
                               f"siblings of the main stream; a nested fork crashes hipStreamEndCapture)")
        if self._n == len(self._forks):
            self._forks.append(torch.cuda.Event())
        ev = self._forks[self._n]
        self._n += 1
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            Branch._depth += 1
            try:
                fn()
            finally:
                Branch._depth -= 1
            self._join.record(self.stream)

    def join(self):
        if self.on and self._n:
            torch.cuda.current_stream().wait_event(self._join)
        self._n = 0


def _colsum(buf: Buffers, tag: str, x2d: torch.Tensor, out_vec: torch.Tensor, accumulate: bool):
    """out_vec[N] (+)= sum over all rows of x2d[M,N] -- one launch for short M, two for long M."""
    M, N = x2d.shape
    if M <= 512:
        hip.segment_colsum(x2d, out_vec.view(1, N), seg=M, mode=0, accumulate=accumulate)
    else:
        chunk = 128
        part = buf.get(tag + ".colsum", ((M + chunk - 1) // chunk, N), torch.float32)
        hip.segment_colsum(x2d, part, seg=chunk, mode=0)
        hip.segment_colsum(part, out_vec.view(1, N), seg=part.shape[0], mode=0, accumulate=accumulate)


def _wgrad_group(buf: Buffers, problems, defer: list, later: Optional[list] = None, time_bwd=None):
    """problems: [(dz, x, dw, ws_tag[, dbias])] -- the slabs of every problem in ONE launch when they all qualify for the ring
    kernel (one by one otherwise); appends (workspace, nslab, dw) to `defer` for the step's single reduction.
    dbias (optional fp32 [N] gradient of the layer's bias) with `later` given: the same launch leaves the bias gradient's
    per-slice partial sums (one extra MFMA per row tile, csrc/gemm.hip) and (partial rows, rows, dbias) goes to `later`;
    returns the problems whose bias gradient still has to be summed by the caller (not in the grouped launch)."""
    items = []
    for pr in problems:
        dz, x, dw, tag = pr[:4]
        M, N = dz.shape
        K = x.shape[1]
        if K % 4 != 0 or dw.stride(0) % 4 != 0:
            raise hip.HipError("_wgrad_group: gradient rows must be 16-byte aligned")
        items.append((dz, x, buf.bytes(tag, int(hip.lib().ib_linear_wgrad_slabs_workspace(M, N, K)))))
    # only problems whose operand rows are 16-byte aligned qualify for the grouped (ring) launch: the others -- e.g. the
    # 30-column output gradient of Groundlink's last layer -- are issued on their own instead of dragging the group down
    def aligned16(t):
        return t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0
    grp = [i for i, (dz, x, ws) in enumerate(items) if aligned16(dz) and aligned16(x) and dz.dtype == torch.bfloat16]
    ns = [None] * len(items)
    parts = [None] * len(items)
    if 1 < len(grp) <= 6:
        if later is not None and not TU.no_wgrad_bias:
            for i in grp:
                if len(problems[i]) > 4 and problems[i][4] is not None:
                    parts[i] = buf.get(problems[i][3] + ".bpart", (32, items[i][0].shape[1]), torch.float32)
        got = None
        if time_bwd is not None and len(grp) == len(items) and not any(p_ is not None for p_ in parts):
            # the time-MLP backward's workgroups ride in this launch (time_bwd = [operands, consumed flag])
            got = hip.linear_wgrad_slabs_multi([items[i] for i in grp], time_bwd=time_bwd[0])
            time_bwd[1] = got is not None
        if got is None:
            got = hip.linear_wgrad_slabs_multi([items[i] for i in grp], bias_parts=[parts[i] for i in grp])
        if got is not None:
            for i, n in zip(grp, got):
                ns[i] = n
        else:
            parts = [None] * len(items)
    for i, (dz, x, ws) in enumerate(items):
        if ns[i] is None:
            ns[i] = hip.linear_wgrad_slabs(dz, x, ws)
            parts[i] = None
    todo = []
    for i, ((dz, x, ws), n, pr) in enumerate(zip(items, ns, problems)):
        defer.append((ws, n, pr[2]))
        if len(pr) > 4 and pr[4] is not None:
            if parts[i] is not None:
                later.append((parts[i][:n], n, pr[4]))
            else:
                todo.append(pr)
    return todo


def _wgrad(buf: Buffers, dz: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, accumulate: bool, ws_tag: str = "wgrad.ws",
           defer: Optional[list] = None):
    """`ws_tag`: launches that may run concurrently (different branches) must not share a slab workspace.
    `defer`: only write the split-M slabs now and append (workspace, nslab, dw) -- the caller sums every deferred
    gradient with ONE hip.slab_reduce_multi launch (each kernel boundary of a captured step costs ~4.5 us)."""
    M, N = dz.shape
    K = x.shape[1]
    if defer is not None and K % 4 == 0 and dw.stride(0) % 4 == 0:
        ws = buf.bytes(ws_tag, int(hip.lib().ib_linear_wgrad_slabs_workspace(M, N, K)))
        defer.append((ws, hip.linear_wgrad_slabs(dz, x, ws), dw))
        return
    ws = buf.bytes(ws_tag, hip.linear_wgrad_workspace_bytes(M, N, K))
    hip.linear_wgrad(dz, x, dw, ws, accumulate=accumulate)


def _as_dtype(buf: Buffers, tag: str, t32: torch.Tensor, dtype) -> torch.Tensor:
    """fp32 reduction result -> GEMM operand in the compute dtype (no-op in parity mode)."""
    if dtype == torch.float32:
        return t32
    o = buf.get(tag + ".cast", t32.shape, dtype)
    hip.cast(t32, o)
    return o


# ------------------------------------------------------------------------------------------------
class DenseStackPlan:
    """[Dropout] [BatchNorm1d] Linear act ... [Dropout] [BatchNorm1d] Linear over [M, K0] rows: the reference's layer
    build, src/models/FeedForwardRegressionBaseline.py:65-77 (`if dropout: Dropout(p)`, `if batchnorm: BatchNorm1d(h0)`,
    Linear, activation on all but the last).

    Dropout = the counter-hash mask kernel keyed on (seed + layer, step, element), regenerated in the backward instead of
    stored (as in GroundlinkPlan).  BatchNorm1d = csrc/batchnorm.hip: batch statistics + running-statistics update in
    train mode, running statistics in eval mode.  With a BatchNorm in front of Linear i the activation derivative of
    layer i-1 cannot ride in the dgrad epilogue (the normalisation sits between them): it is multiplied in by the
    BatchNorm backward pass instead; the dropout mask (elementwise, commutes) follows in place."""

    def __init__(self, names: Sequence[Tuple[str, str]], activation: str, dtype, device, tag="ff",
                 bn_names: Optional[Sequence[Optional[Tuple[str, str]]]] = None, dropout_p: float = 0.0, seed: Optional[int] = None):
        self.names, self.act, self.dtype, self.tag = list(names), activation, dtype, tag
        self.bn = list(bn_names) if bn_names is not None else [None] * len(self.names)
        self.p, self.seed = float(dropout_p), mask_seed(0x2F1) if seed is None else int(seed)
        self.bn_buffers: Dict[str, Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]] = {}   # set by the model
        self.buf = Buffers(device)
        self.saved: List = []
        self.ctx = (False, 0, None)

    def forward(self, x: torch.Tensor, P: ParamSource, out: Optional[torch.Tensor] = None, training: bool = False,
                step: int = 0, step_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
        M = x.shape[0]
        L = len(self.names)
        g, dt, tg = self.buf.get, self.dtype, self.tag
        drop = training and self.p > 0.0
        self.saved = []
        self.ctx = (training, step, step_dev)
        h = x
        for i, (wn, bn) in enumerate(self.names):
            w = P.w(wn)
            last = i == L - 1
            if drop:
                d = g(f"{tg}.drop{i}", h.shape, dt)
                hip.dropout(h, d, self.p, self.seed + i, step, step_dev)
                h = d
            bn_in = None
            if self.bn[i] is not None:
                gn, btn = self.bn[i]
                rm, rv, nbt = self.bn_buffers[gn]
                C = h.shape[1]
                sm, sr = g(f"{tg}.bnm{i}", (C,), torch.float32), g(f"{tg}.bnr{i}", (C,), torch.float32)
                yb = g(f"{tg}.bn{i}", h.shape, dt)
                hip.batchnorm_fwd(h, P.v(gn), P.v(btn), rm, rv, nbt, yb, sm, sr, training)
                bn_in, h = (h, sm, sr), yb
            y = out if (last and out is not None) else g(f"{tg}.y{i}", (M, w.shape[0]), dt)
            z = None
            if not last and self.act == "silu":
                z = g(f"{tg}.z{i}", (M, w.shape[0]), dt)
            hip.linear_fwd(h, w, P.v(bn), y, act="none" if last else self.act, z=z)
            self.saved.append((h, bn_in, y, z))          # Linear input, BatchNorm (input, mean, rstd), output, pre-activation
            h = y
        return h

    def ready_order(self) -> List[str]:
        o = []
        for i in range(len(self.names) - 1, -1, -1):
            o += list(self.names[i])
            if self.bn[i] is not None:
                o += list(self.bn[i])
        return o

    def backward(self, dout: torch.Tensor, P: ParamSource, accumulate=False, need_dx=False):
        L = len(self.names)
        g, dt, tg = self.buf.get, self.dtype, self.tag
        training, step, step_dev = self.ctx
        drop = training and self.p > 0.0
        dz = dout
        dx = None
        for i in range(L - 1, -1, -1):
            wn, bn = self.names[i]
            lin_in, bn_in, _, _ = self.saved[i]
            # short reductions (a batch of a few hundred rows): weight and bias gradient from one launch
            # (bf16 up to 1024 rows; fp32 up to 256 -- the reference's own batch sizes: csrc/gemm_f32_small.hip)
            if not (dz.shape[0] <= 1024 and hip.linear_wgrad_bias(dz, lin_in, P.g(wn), P.g(bn), accumulate)):
                _wgrad(self.buf, dz, lin_in, P.g(wn), accumulate)
                _colsum(self.buf, f"{tg}.b{i}", dz, P.g(bn), accumulate)
            P.ready(wn)
            P.ready(bn)
            below = i > 0 or need_dx
            if not below and bn_in is None:
                continue
            # derivative of the activation below this layer (layer i-1's), from its output (its pre-activation for silu)
            act_below, aux = "none", None
            if i > 0:
                _, _, y_prev, z_prev = self.saved[i - 1]
                act_below, aux = self.act, (z_prev if z_prev is not None else y_prev)
            if bn_in is not None:
                xin, sm, sr = bn_in
                gn, btn = self.bn[i]
                db = g(f"{tg}.db{i}", lin_in.shape, dt)
                hip.linear_dgrad(dz, P.w(wn), db)
                cur = g(f"{tg}.dz{i - 1}" if i > 0 else f"{tg}.dx", lin_in.shape, dt) if below else None
                hip.batchnorm_bwd(db, xin, P.v(gn), sm, sr, cur, P.g(gn), P.g(btn), training, accumulate=accumulate,
                                  act_below=act_below if below else "none", aux=aux if below else None)
                P.ready(gn)
                P.ready(btn)
            else:
                cur = g(f"{tg}.dz{i - 1}" if i > 0 else f"{tg}.dx", lin_in.shape, dt)
                hip.linear_dgrad(dz, P.w(wn), cur, act_below=act_below, aux=aux)
            if below and drop:
                hip.dropout(cur, cur, self.p, self.seed + i, step, step_dev)      # the forward's mask, regenerated
            if i > 0:
                dz = cur
            else:
                dx = cur
        return dx


class GroundlinkPlan:
    """Groundlink (src/models/Groundlink.py:34-62): four Conv1d(k=7, replicate) + ELU over the frames of a window, then per
    frame [Dropout, Linear, ELU] x2 and [Dropout, Linear(no bias)].  A convolution = im2col (clamped-frame gather) + the
    fused Linear+bias+ELU GEMM over `weight.view(C_out, C_in*k)`; its backward = wgrad over the saved im2col, dgrad into
    the im2col gradient, col2im with the ELU derivative of the layer below fused.  Dropout is a counter-based mask
    (seed, step, element) regenerated in the backward.  The first convolution's reduction (177 x 7 = 1239) is not a
    multiple of 8: its im2col rows are pitched to 1240 and the forward GEMM reads a zero-padded copy of the weight, so
    every operand piece stays a 16-byte load."""

    CONV = (1, 4, 7, 10)
    FC = (2, 5)

    def __init__(self, output_data_format: str, dtype, device, fc_dropout: float = 0.2, k: int = 7, seed: Optional[int] = None):
        self.fmt, self.dtype, self.p, self.k = output_data_format, dtype, float(fc_dropout), k
        self.seed = mask_seed(0x1B3) if seed is None else int(seed)
        self.buf = Buffers(device)
        self.ctx = None
        self.fuse_reduce_into_optimizer = False         # set by HipTrainer for single-GPU steps
        self.pending_sources = None

    def _conv_weight(self, li: int, w: torch.Tensor) -> torch.Tensor:
        co, ci, k = w.shape
        K = ci * k
        Kp = (K + 7) // 8 * 8
        if Kp == K:
            return w.view(co, K)
        wp = self.buf.get(f"gl.wp{li}", (co, Kp), self.dtype)
        wp[:, K:].zero_()
        wp[:, :K].copy_(w.view(co, K))
        return wp

    def forward(self, x: torch.Tensor, P: ParamSource, out: Optional[torch.Tensor] = None, training: bool = False,
                step: int = 0, step_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: [N, F*C0] (frame-major concat of the input keys = channels-last rows [N*F, C0]) -> [N, F', 30]"""
        g, dt, k = self.buf.get, self.dtype, self.k
        c0 = P.w("cnn.1.weight").shape[1]
        N, F = x.shape[0], x.shape[1] // c0
        if F * c0 != x.shape[1]:
            raise hip.HipError(f"groundlink: input width {x.shape[1]} is not a multiple of {c0} channels")
        h = x.view(N * F, c0)
        convs = []
        for li, idx in enumerate(self.CONV):
            w = P.w(f"cnn.{idx}.weight")
            co, ci, _ = w.shape
            w2 = self._conv_weight(li, w)
            col = g(f"gl.col{li}", (N * F, w2.shape[1]), dt)
            hip.im2col_replicate(h, col, N, F, k)
            y = g(f"gl.y{li}", (N * F, co), dt)
            hip.linear_fwd(col, w2, P.v(f"cnn.{idx}.bias"), y, act="elu")
            convs.append((col, y))
            h = y
        last = self.fmt != "all_frames"
        R = N if last else N * F
        C = h.shape[1]
        if last:                                                   # Groundlink.py:148: only the last frame feeds fc
            hin = g("gl.hin", (N, C), dt)
            hin.copy_(h.view(N, F, C)[:, -1, :])
        else:
            hin = h
        drop = training and self.p > 0.0
        fcs = []
        a = hin
        for j, idx in enumerate(self.FC):
            d = a
            if drop:
                d = g(f"gl.d{j}", (R, C), dt)
                hip.dropout(a, d, self.p, self.seed + j, step, step_dev)
            y = g(f"gl.a{j}", (R, C), dt)
            hip.linear_fwd(d, P.w(f"fc.{idx}.weight"), P.v(f"fc.{idx}.bias"), y, act="elu")
            fcs.append((d, y))
            a = y
        d = a
        if drop:
            d = g("gl.d2", (R, C), dt)
            hip.dropout(a, d, self.p, self.seed + 2, step, step_dev)
        Fo = 1 if last else F
        out = out if out is not None else g("gl.out", (N, Fo, 30), dt)
        hip.linear_fwd(d, P.w("fc.8.weight"), None, out.view(R, 30))
        self.ctx = (convs, hin, fcs, d, N, F, drop, step, step_dev)
        return out

    def ready_order(self) -> List[str]:
        o = ["fc.8.weight"]
        for idx in reversed(self.FC):
            o += [f"fc.{idx}.weight", f"fc.{idx}.bias"]
        for idx in reversed(self.CONV):
            o += [f"cnn.{idx}.weight", f"cnn.{idx}.bias"]
        return o

    def _dbias(self, tag: str, dz: torch.Tensor, dst: torch.Tensor, accumulate: bool, later: list):
        """d bias = column sums of dz: short matrices in one launch; long ones leave per-128-row partial sums for the
        step's final reduction (the optimizer itself on one GPU)"""
        M, Nc = dz.shape
        if M <= 512:
            hip.segment_colsum(dz, dst.view(1, Nc), seg=M, mode=0, accumulate=accumulate)
            return
        part = self.buf.get(tag + ".colsum", ((M + 127) // 128, Nc), torch.float32)
        hip.segment_colsum(dz, part, seg=128, mode=0)
        later.append((part, part.shape[0], dst))

    def backward(self, dout: torch.Tensor, P: ParamSource, accumulate=False):
        """Every dgrad first (the chain of dependent launches), the weight gradients after it in grouped launches whose
        split-M slabs -- like the bias partial sums -- are reduced once: by ib_optim_step_sources when the trainer fuses
        the reduction into the optimizer (one GPU), by one ib_slab_reduce_multi launch otherwise."""
        convs, hin, fcs, d_last, N, F, drop, step, step_dev = self.ctx
        g, dt, k = self.buf.get, self.dtype, self.k
        last = self.fmt != "all_frames"
        R = N if last else N * F
        C = d_last.shape[1]
        dz = dout.reshape(R, 30)
        wg, later = [], []                 # (dz, x, dw, workspace tag) ; (partial sums, rows, bias gradient)
        # fc.8 (no bias), then back through [ELU, Dropout] of fc.5 and fc.2.  The ELU derivative is fused into the dgrad
        # epilogue; it is elementwise, so it commutes with the dropout mask applied right after.
        wg.append((dz, d_last, P.g("fc.8.weight"), "gl.ws8"))
        wname = "fc.8.weight"
        for j in range(len(self.FC) - 1, -1, -1):
            idx = self.FC[j]
            d_in, y = fcs[j]
            dy = g(f"gl.da{j}", (R, C), dt)
            hip.linear_dgrad(dz, P.w(wname), dy, act_below="elu", aux=y)
            if drop:
                hip.dropout(dy, dy, self.p, self.seed + j + 1, step, step_dev)
            dz = dy
            wname = f"fc.{idx}.weight"
            wg.append((dz, d_in, P.g(wname), f"gl.wsf{j}"))
            self._dbias(f"gl.bf{j}", dz, P.g(f"fc.{idx}.bias"), accumulate, later)
        # into the last convolution's output: (dz W_fc2) x ELU'(y3) [x dropout mask 0]
        dh = g("gl.dh", (R, C), dt)
        hip.linear_dgrad(dz, P.w(wname), dh, act_below="elu", aux=hin)
        if drop:
            hip.dropout(dh, dh, self.p, self.seed, step, step_dev)
        if last:                                                   # frames before the last one get no gradient from fc
            dz = g("gl.dzc3", (N * F, C), dt)
            dz.zero_()
            dz.view(N, F, C)[:, -1, :].copy_(dh)
        else:
            dz = dh
        for li in range(len(self.CONV) - 1, -1, -1):
            idx = self.CONV[li]
            col, y = convs[li]
            w = P.w(f"cnn.{idx}.weight")
            co, ci, _ = w.shape
            K = ci * k
            wg.append((dz, col[:, :K], P.g(f"cnn.{idx}.weight").view(co, K), f"gl.wsc{li}"))
            self._dbias(f"gl.bc{li}", dz, P.g(f"cnn.{idx}.bias"), accumulate, later)
            if li > 0:
                dcol = g(f"gl.dcol{li}", (N * F, K), dt)
                hip.linear_dgrad(dz, w.view(co, K), dcol)
                nxt = g(f"gl.dzc{li - 1}", (N * F, ci), dt)
                hip.col2im_replicate(dcol, nxt, N, F, k, act="elu", aux=convs[li - 1][1])
                dz = nxt
        # ---- weight gradients: slabs of up to 6 problems per launch; odd reductions (177 x 7) take the direct kernel
        defer, group = [], []
        for prob in wg:
            if prob[1].shape[1] % 4 == 0 and prob[2].stride(0) % 4 == 0:
                group.append(prob)
            else:
                _wgrad(self.buf, prob[0], prob[1], prob[2], accumulate, ws_tag=prob[3])
        for a in range(0, len(group), 6):
            _wgrad_group(self.buf, group[a:a + 6], defer)
        self.pending_sources = None
        if self.fuse_reduce_into_optimizer and not accumulate:
            segs = [(0, part.shape[1], dst, None, 1.0, part, rows) for part, rows, dst in later]
            self.pending_sources = (defer, None, 0, segs)
        else:
            if defer:
                hip.slab_reduce_multi(defer, accumulate=accumulate)
            for part, rows, dst in later:      # ib_colsum_segments sums in the optimizer's order (bitwise the fused path)
                hip.colsum_segments(part, rows, [(0, part.shape[1], dst, None, 1.0)], accumulate=accumulate)
        for name in self.ready_order():
            P.ready(name)
        return None


# ------------------------------------------------------------------------------------------------
class TransformerLayerPlan:
    """Post-norm encoder layer (TransformerBaseline.py:24-38): x=LN1(x+Drop(MHA(x))); x=LN2(x+Drop(W2 relu(W1 x))).

    dropout_p (train mode only): the layer's three dropouts -- on the attention probabilities inside the attention kernels
    (nn.MultiheadAttention(dropout=p), :12-13), dropout1 on the attention block's output (:30), dropout2 on the feedforward
    output (:35).  All masks are counter-based hashes of (seed + site, step, element), regenerated in the backward."""

    def __init__(self, prefix: str, d_model: int, num_heads: int, ffn: int, dtype, device, buf: Optional[Buffers] = None,
                 tag="tl", dropout_p: float = 0.0, seed: Optional[int] = None):
        self.p, self.d, self.h, self.ffn, self.dtype, self.tag = prefix, d_model, num_heads, ffn, dtype, tag
        self.drop_p = float(dropout_p)
        # per plan instance: the tag (it carries the layer index) enters the seed through a CRC of the whole string, so
        # stacked layers draw different masks ('tl12' and 'tl21' too: a character sum does not see digit order)
        self.seed = mask_seed(0x3A7 + 16 * (zlib.crc32(tag.encode()) & 0xFFFFFF)) if seed is None else int(seed)
        self.buf = buf if buf is not None else Buffers(device)
        self.ctx = None
        # the four weight-gradient GEMMs (+ bias sums) hang off the critical dgrad / LayerNorm / attention chain.  One GPU,
        # large batches: they are ONE grouped launch per layer (csrc/gemm_tn.hip), issued on a side stream of the layer so the
        # next layer's chain starts beside it (B = 256, T = 50, same box: 2.45 ms inline, 2.41 ms forked; joined once at the
        # end of the whole backward).  Everything else runs them inline: as separate launches on a side stream they were
        # slower (3.059 vs 3.123 ms with the round-1 kernels), and a data-parallel layer joins right away for its bucket.
        # IB_LAYER_BRANCH=1 forces the side stream for every shape, IB_NO_LAYER_BRANCH=1 forces inline.
        self.branch = Branch(device, enabled=not TU.no_layer_branch, name="layer")
        self._always_fork = bool(TU.layer_branch)
        # transposed bf16 copies of the four weight matrices: the dgrad GEMMs of large batches read them k-contiguously
        # (ib_linear_dgrad_wt -> the 256 x 128 LDS-DMA kernel of csrc/gemm_nt.hip); refreshed once per forward
        self._wt: Dict[str, torch.Tensor] = {}
        self._wt_fresh = False
        # set by a parent plan that stacks layers: the layer ABOVE this one (its in-projection rides behind this layer's
        # LayerNorm2 in the fused forward launch) and the layer BELOW (this layer's in-projection dgrad + residual addend
        # ride in front of that layer's fused backward launch) -- csrc/ffn_chain.hip's QKV tail / head
        self.qkv_tail_for: Optional["TransformerLayerPlan"] = None
        self.qkv_dgrad_below: Optional["TransformerLayerPlan"] = None
        self.own_wt = True           # a parent plan refreshes the copies of all its layers in ONE launch instead
        self.join_on_exit = True     # a parent plan sets this False and joins all layers once at the end
        self.inference = False       # forward only (DDIM sampler): Linear + residual + LayerNorm fused, nothing saved
        # set by a parent plan whose trainer lets the optimizer sum partial gradients (one GPU): weight-gradient slabs go
        # to `defer` [(workspace, nslab, dw)], bias / LayerNorm partial sums to `later` [(partial rows, rows, gradient)]
        self.defer: Optional[list] = None
        self.later: Optional[list] = None
        self.flush_on_exit = False   # data parallel with overlapped all-reduces: join the side stream and let the trainer
                                     # launch the completed buckets at the end of every layer's backward
        # data parallel, set by a parent plan for every layer but the one whose backward runs last: the layer's grouped
        # weight-gradient launch + its reduction are NOT issued by backward(); they are handed back (take_lagged) and the
        # parent forks them beside the NEXT layer's backward, joins them at that layer's end and only then reports the
        # gradients ready and flushes.  A captured graph segment must end with every side stream joined, so without the lag
        # each layer's 100-us launch sat on the critical path once per bucket.
        self.lag_group = False
        self.tail_done = self.attn_tail_done = False
        self.split_tail = False       # set by the parent on the layer whose backward runs LAST (see backward())
        self.infer_packed = False     # set by the parent's prepare_inference: packed_image() holds the frozen weights
        self.parent_flushes = False  # set with lag_group on ALL layers of such a parent: gradients are reported to the parent
                                     # (take_lagged), which keeps the ready order and does the flushes
        self._lagged = None

    def branches(self) -> List["Branch"]:
        return [self.branch]

    def take_lagged(self):
        """(closure issuing the layer's grouped weight-gradient launch + reduction, names to report ready) or None"""
        lg, self._lagged = self._lagged, None
        return lg

    WT_NAMES = ("feedforward.2.weight", "feedforward.0.weight", "multihead_attention.out_proj.weight",
                "multihead_attention.in_proj_weight")

    def wt_pairs(self, P: ParamSource, M: int):
        """[(weight, its transposed copy)] to refresh before a training forward, or [] when the large-M dgrad path does not
        apply (fp32 parity mode, inference, small token counts)"""
        self._wt_fresh = False
        if self.dtype != torch.bfloat16 or self.inference or M < 4096 or TU.no_nt:
            return []
        pairs = []
        skip = ("feedforward.2.weight", "feedforward.0.weight", "multihead_attention.out_proj.weight") if self.ffn_fused(M) else ()
        if skip and self.qkv_dgrad_below is not None and self.qkv_dgrad_below.ffn_fused(M) \
                and not TU.no_qkv_fuse:
            skip = skip + ("multihead_attention.in_proj_weight",)
        for n in self.WT_NAMES:
            if n in skip:          # the fused feed-forward sublayer streams its own packed images (ffn_pack_item)
                continue
            w = P.w(self.p + n)
            wt = self.buf.get(self.tag + ".wt." + n, (w.shape[1], w.shape[0]), self.dtype)
            self._wt[n] = wt
            pairs.append((w, wt))
        self._wt_fresh = True
        return pairs

    def ffn_fused(self, M: int) -> bool:
        """everything of the layer behind the attention core -- out-projection + residual + LayerNorm1, then the feed-forward
        sublayer Linear + ReLU + Linear + residual + LayerNorm2 (TransformerBaseline.py:12-19,29-36) -- and its backward as
        ONE launch each (csrc/ffn_chain.hip): bf16 training at chip-filling token counts, d = 512, a hidden width of whole
        512-column chunks, no dropout between the Linears and the residuals"""
        return (self.dtype == torch.bfloat16 and not self.inference and M >= 4096 and self.drop_p == 0.0
                and not TU.no_ffn_chain and not TU.no_nt
                and hip.ffn_chain_supported(self.d, self.ffn))

    def attn_T(self, M: int, T: int) -> int:
        """T when the attention core rides INSIDE the layer's fused launches (round 5; csrc/ffn_chain.h): the panels are then
        exactly one window of T frames (16 <= T <= 64) whose eight 64-column heads belong to the eight waves -- forward
        attention of the layer above behind the QKV tail, this layer's attention backward + in-projection dgrad behind the
        out-projection's dgrad: the whole layer is one launch per direction.  0: the token-count panels, separate
        attention launches."""
        if (self.ffn_fused(M) and not TU.no_attn_fuse and not TU.no_qkv_fuse and self.d == 512 and self.h == 8
                and 16 <= T <= 64 and M % T == 0 and hip.ffn_chain_workgroups(M, self.d, self.ffn, T) > 0):
            # a panel kernel's time per workgroup is set by the weights it streams, whatever its row count: one-window panels
            # much shorter than the token-count panels (many short windows: B = 2048, T = 16 -> 2048 workgroups instead of
            # 512) multiply the rounds of workgroups; there the separate attention launches win
            natural = min(64, max(16, -(-M // 256)))
            if 4 * T >= 3 * natural:
                return T
        return 0

    def tail_active(self, M: int, training: bool = False) -> bool:
        """this layer's fused forward launch also computes the in-projection of the layer above (both packed images are
        addressed with one chunk count: the two layers must have the same hidden width)"""
        nxt = self.qkv_tail_for
        return (nxt is not None and self.ffn_fused(M) and nxt.ffn_fused(M) and not TU.no_qkv_fuse and nxt.ffn == self.ffn
                and not (training and (self.drop_p > 0.0 or nxt.drop_p > 0.0)))

    def attn_tail_active(self, M: int, T: int, training: bool = False) -> bool:
        """... and that layer's attention core behind it"""
        nxt = self.qkv_tail_for
        return bool(self.tail_active(M, training) and self.attn_T(M, T) and nxt.attn_T(M, T) and nxt.h == self.h)

    def head_active(self, M: int) -> bool:
        """this layer's in-projection dgrad (+ residual addend) is computed by the fused backward launch of the layer below"""
        low = self.qkv_dgrad_below
        return (low is not None and self.ffn_fused(M) and low.ffn_fused(M) and not TU.no_qkv_fuse and low.ffn == self.ffn
                and getattr(self, "_ffn_fused", False) and getattr(low, "_ffn_fused", False)
                and not getattr(self, "_att_T", 0))

    def ffn_pack_item(self, P: ParamSource, M: int):
        """(feedforward.0.weight, feedforward.2.weight, this layer's packed image) for ib_ffn_chain_pack -- refreshed once
        per training step, all layers of a parent plan in ONE launch -- or None"""
        if not self.ffn_fused(M):
            return None
        pk = self.buf.get(self.tag + ".ffnpk", (hip.ffn_chain_packed_elems(self.d, self.ffn),), self.dtype)
        return (P.w(self.p + "feedforward.0.weight"), P.w(self.p + "feedforward.2.weight"), pk,
                P.w(self.p + "multihead_attention.out_proj.weight"), P.w(self.p + "multihead_attention.in_proj_weight"))

    def qkv_buffer(self, B: int, T: int) -> torch.Tensor:
        return self.buf.get(self.tag + ".qkv", (B, T, 3 * self.d), self.dtype)

    def attn_buffers(self, B: int, T: int):
        return (self.buf.get(self.tag + ".attn", (B, T, self.d), self.dtype),
                self.buf.get(self.tag + ".lse", (B, self.h, T), torch.float32))

    def packed_image(self) -> torch.Tensor:
        return self.buf.get(self.tag + ".ffnpk", (hip.ffn_chain_packed_elems(self.d, self.ffn),), self.dtype)

    def _dgrad(self, P: ParamSource, dz, wname: str, dx, act_below="none", aux=None, addend=None):
        wt = self._wt.get(wname) if self._wt_fresh else None
        if wt is not None and hip.linear_dgrad_wt(dz, wt, dx, act_below=act_below, aux=aux, addend=addend):
            return
        hip.linear_dgrad(dz, P.w(self.p + wname), dx, act_below=act_below, aux=aux, addend=addend)

    def forward(self, x3: torch.Tensor, P: ParamSource, out: Optional[torch.Tensor] = None, training: bool = False,
                step: int = 0, step_dev: Optional[torch.Tensor] = None, qkv_ready: bool = False,
                attn_ready: bool = False) -> torch.Tensor:
        """qkv_ready: the layer below already wrote this layer's in-projection into `qkv_buffer` (its fused launch's tail);
        attn_ready: ... and this layer's attention output + row log-sum-exp (`attn_buffers`) behind it"""
        B, T, d = x3.shape
        M = B * T
        g, dt, p, tg = self.buf.get, self.dtype, self.p, self.tag
        drop = None
        if training and self.drop_p > 0.0:
            if self.inference:
                raise hip.HipError("TransformerLayerPlan: inference mode with training=True")
            drop = (self.drop_p, self.seed, step, step_dev)
        ffn_fused = self.ffn_fused(M) and drop is None
        if self.own_wt:
            pairs = self.wt_pairs(P, M)
            if pairs:
                hip.transpose_multi(pairs)
            if ffn_fused:
                hip.ffn_chain_pack([self.ffn_pack_item(P, M)])
        x = x3.view(M, d)
        qkv = g(tg + ".qkv", (B, T, 3 * d), dt)
        if not qkv_ready:
            nc = self.ffn // 512
            if not (self.inference and self.infer_packed and d == 512 and dt == torch.bfloat16
                    and not TU.no_qkv_panel and M <= TU.qkv_panel_max_m
                    and hip.linear_panel_fwd(x, self.packed_image()[(4 * nc + 2) * 512 * 512:(4 * nc + 5) * 512 * 512],
                                             P.v(p + "multihead_attention.in_proj_bias"), qkv.view(M, 3 * d))):
                hip.linear_fwd(x, P.w(p + "multihead_attention.in_proj_weight"), P.v(p + "multihead_attention.in_proj_bias"),
                               qkv.view(M, 3 * d))
        attn, lse = self.attn_buffers(B, T)
        if not attn_ready:
            hip.attention_fwd(qkv, attn, lse, self.h, drop=drop)
        x2 = out if out is not None else g(tg + ".x2", (B, T, d), dt)
        self.tail_done = self.attn_tail_done = False          # read by a parent plan: what this launch did for the layer above
        if (self.inference and self.infer_packed and d == 512 and dt == torch.bfloat16 and M >= TU.infer_chain_min_m
                and not TU.no_infer_chain and x2.is_contiguous() and hip.ffn_chain_supported(d, self.ffn)):
            # frozen weights, more rows than the row-panel kernels take (DDIM at B = 256: 51200 rows): everything behind the
            # attention core -- and the next layer's in-projection -- in ONE launch over 64-row panels, the training
            # launch's form that saves nothing (csrc/ffn_chain.hip, INFER).  Round 5: 8 per-op launches per layer before.
            nxt = self.qkv_tail_for
            tail = None
            if nxt is not None and nxt.infer_packed and nxt.ffn == self.ffn and nxt.d == d and not TU.no_qkv_fuse:
                tail = (nxt.packed_image(), P.v(nxt.p + "multihead_attention.in_proj_bias"), nxt.qkv_buffer(B, T).view(M, 3 * d))
            if hip.ffn_chain_fwd_infer(x, self.packed_image(), P.v(p + "feedforward.0.bias"), P.v(p + "feedforward.2.bias"),
                                       P.v(p + "norm2.weight"), P.v(p + "norm2.bias"), x2.view(M, d), attn.view(M, d),
                                       P.v(p + "multihead_attention.out_proj.bias"), P.v(p + "norm1.weight"),
                                       P.v(p + "norm1.bias"), qkv_next=tail):
                self.tail_done = tail is not None
                self.ctx = None
                return x2
        x1 = g(tg + ".x1", (M, d), dt)
        f1 = g(tg + ".f1", (M, self.ffn), dt)
        fuse = self.inference and dt == torch.bfloat16 and not TU.no_linear_ln

        def lin_ln(inp, wname, bname, nname, res, y, wtag):
            """Linear -> +res -> LayerNorm as one K-split GEMM + fused reduction (small M); False = not applicable"""
            w = P.w(p + wname)
            # swept on the sampler (T = 200): K = 2048 fused wins up to the NT kernel's territory (M = 3200: 43.6 -> 30.5 us);
            # K = 512 (two short kernels either way) wins only while the GEMM is far from filling the chip (M <= 2048:
            # B = 2 / 4 / 8 +4 / +6 / +2 % steps/s; M = 3200: -6 % on the ring kernel).  Round 3: from 640 rows the GEMM of
            # the fused form is the 256 x 128 kernel in split-K form (ib_gemm_nt_splitk): K = 2048 31.2 -> 24.0 us at
            # M = 3200, B = 4 / 8 / 16 +7.5 / +8 / +7 % steps/s; K = 512 fused now pays up to M < 4096 too (+1-2 %)
            if w.shape[1] < 512 or (w.shape[1] < 1024 and M > TU.linln_k512_max_m):
                return False
            if M >= TU.linln_max_m and not TU.no_nt:
                return False           # large batches fill the chip without a K split: the 256 x 128 NT kernel + LayerNorm
                                       # (M = 51200: 120 + 30 us against 235 us fused)
            ws = self.buf.bytes(tg + wtag, int(hip.lib().ib_linear_ln_fwd_workspace(M, w.shape[0], w.shape[1])))
            return hip.linear_ln_fwd(inp, w, P.v(p + bname), res, P.v(p + nname + ".weight"), P.v(p + nname + ".bias"), y, ws)

        a = m1 = r1 = f2 = m2 = r2 = None
        self._ffn_fused = ffn_fused
        # one-window panels in BOTH directions of this layer (the ReLU bit words are laid out per panel)
        Ta = self._att_T = self.attn_T(M, T) if ffn_fused else 0
        if ffn_fused:
            # a / f2 of the context = the LayerNorm INPUTS x + o and x1 + f2 the kernel stores (its backward normalises
            # those; no `res`)
            a, f2 = g(tg + ".s1", (M, d), dt), g(tg + ".s2", (M, d), dt)
            m1, r1 = g(tg + ".m1", (M,), torch.float32), g(tg + ".r1", (M,), torch.float32)
            m2, r2 = g(tg + ".m2", (M,), torch.float32), g(tg + ".r2", (M,), torch.float32)
            mask = self.buf.get(tg + ".ffnmask", (hip.ffn_chain_mask_bytes(M, d, self.ffn, Ta),), torch.uint8)
            nxt = self.qkv_tail_for
            att_next = None
            if Ta and self.attn_tail_active(M, T, training):
                na, nl = nxt.attn_buffers(B, T)
                att_next = (na.view(M, d), nl, Ta)
            hip.ffn_chain_fwd(x, self.buf.get(tg + ".ffnpk", (hip.ffn_chain_packed_elems(d, self.ffn),), dt),
                              P.v(p + "feedforward.0.bias"), P.v(p + "feedforward.2.bias"), P.v(p + "norm2.weight"),
                              P.v(p + "norm2.bias"), f1, f2, x2.view(M, d), m2, r2, mask,
                              attn_out=(attn.view(M, d), P.v(p + "multihead_attention.out_proj.bias"),
                                        P.v(p + "norm1.weight"), P.v(p + "norm1.bias"), a, x1, m1, r1),
                              qkv_next=None if not self.tail_active(M, training) else
                              (nxt.packed_image(), P.v(nxt.p + "multihead_attention.in_proj_bias"),
                               nxt.qkv_buffer(B, T).view(M, 3 * d)),
                              attn_next=att_next, panel_T=Ta)
            self.tail_done = self.tail_active(M, training)
            self.attn_tail_done = att_next is not None
            self.ctx = (x, qkv, attn, lse, a, x1, m1, r1, f1, f2, m2, r2, B, T, drop)
            return x2
        def panel_ln():
            """out-projection + residual + LayerNorm1 as one launch over row panels (sampler, frozen packed weights)"""
            if not (fuse and self.infer_packed and d == 512 and hip.linear_ln_panel_ok(M, d, d)
                    and M <= TU.linln_panel_max_m):
                return False
            nc = self.ffn // 512
            wo_img = self.packed_image()[4 * nc * 512 * 512:(4 * nc + 1) * 512 * 512]
            return hip.linear_ln_panel_fwd(attn.view(M, d), wo_img, P.v(p + "multihead_attention.out_proj.bias"), x,
                                           P.v(p + "norm1.weight"), P.v(p + "norm1.bias"), x1)
        if panel_ln():
            pass
        elif not (fuse and lin_ln(attn.view(M, d), "multihead_attention.out_proj.weight",
                                  "multihead_attention.out_proj.bias", "norm1", x, x1, ".lnws1")):
            a = g(tg + ".a", (M, d), dt)
            hip.linear_fwd(attn.view(M, d), P.w(p + "multihead_attention.out_proj.weight"),
                           P.v(p + "multihead_attention.out_proj.bias"), a)
            if drop:           # dropout1, in place: only the dropped block output is read again (LayerNorm input)
                hip.dropout(a, a, self.drop_p, self.seed + 1, step, step_dev)
            m1, r1 = g(tg + ".m1", (M,), torch.float32), g(tg + ".r1", (M,), torch.float32)
            hip.layernorm_fwd(a, P.v(p + "norm1.weight"), P.v(p + "norm1.bias"), x1, m1, r1, res=x)
        if (fuse and self.infer_packed and d == 512 and not TU.no_ffn_infer
                and hip.ffn_infer_panels(M, d, self.ffn) and x2.is_contiguous()
                and M <= TU.ffn_infer_max_m):
            # the feed-forward sublayer: a panel of rows is shared by the workgroups of its hidden chunks (both GEMMs, the
            # hidden activation stays in LDS), the slab-reduction LayerNorm launch finishes it (csrc/linln_panel.hip)
            ws = self.buf.bytes(tg + ".ffws", int(hip.lib().ib_ffn_infer_workspace(M, d, self.ffn)))
            hip.ffn_infer_fwd(x1, self.packed_image(), P.v(p + "feedforward.0.bias"), P.v(p + "feedforward.2.bias"),
                              P.v(p + "norm2.weight"), P.v(p + "norm2.bias"), x2.view(M, d), ws)
            self.ctx = None
            return x2
        hip.linear_fwd(x1, P.w(p + "feedforward.0.weight"), P.v(p + "feedforward.0.bias"), f1, act="relu")
        if not (fuse and lin_ln(f1, "feedforward.2.weight", "feedforward.2.bias", "norm2", x1, x2.view(M, d), ".lnws2")):
            f2 = g(tg + ".f2", (M, d), dt)
            hip.linear_fwd(f1, P.w(p + "feedforward.2.weight"), P.v(p + "feedforward.2.bias"), f2)
            if drop:           # dropout2
                hip.dropout(f2, f2, self.drop_p, self.seed + 2, step, step_dev)
            m2, r2 = g(tg + ".m2", (M,), torch.float32), g(tg + ".r2", (M,), torch.float32)
            hip.layernorm_fwd(f2, P.v(p + "norm2.weight"), P.v(p + "norm2.bias"), x2.view(M, d), m2, r2, res=x1)
        self.ctx = (x, qkv, attn, lse, a, x1, m1, r1, f1, f2, m2, r2, B, T, drop)
        return x2

    def ready_order(self) -> List[str]:
        p = self.p
        return [p + n for n in ("norm2.weight", "norm2.bias", "feedforward.2.weight", "feedforward.2.bias",
                                "feedforward.0.weight", "feedforward.0.bias", "norm1.weight", "norm1.bias",
                                "multihead_attention.out_proj.weight", "multihead_attention.out_proj.bias",
                                "multihead_attention.in_proj_weight", "multihead_attention.in_proj_bias")]

    def backward(self, dx2: Optional[torch.Tensor], P: ParamSource, accumulate=False, qkv_head=None) -> Optional[torch.Tensor]:
        """qkv_head = (packed image, dqkv [M, 3 d], ds1 [M, d]) of the layer ABOVE: dx2 (may be None) is then computed inside
        this layer's fused launch.  Returns dx, or None when this layer's own in-projection dgrad is left to the layer below
        (`head_active`; `self.pending_head` then holds the triple to hand down)."""
        x, qkv, attn, lse, a, x1, m1, r1, f1, f2, m2, r2, B, T, drop = self.ctx
        M, d = x.shape
        self.pending_head = None
        g, dt, p, tg = self.buf.get, self.dtype, self.p, self.tag
        lnws = self.buf.bytes("ln.ws", hip.layernorm_bwd_workspace_bytes(M, max(d, 1)))
        defer, later = self.defer, self.later
        # one GPU (slabs go to the optimizer): the layer's four weight-gradient GEMMs are issued as ONE grouped launch at
        # the end of the layer (they share a workgroup budget: fewer, longer slices, half the slab traffic)
        # data parallel / drop-in autograd tier with large batches: the same grouped launch, its slabs and every partial-sum
        # matrix of the layer finished by ONE reduction launch at the end of the layer (before the layer's bucket is
        # all-reduced) instead of a slab reduction per weight and two column-sum launches per bias
        local = defer is None and dt == torch.bfloat16 and M >= 4096 and not accumulate \
            and not TU.no_layer_group
        if local:
            defer, later = [], []
        group = [] if (defer is not None and dt == torch.bfloat16 and not TU.no_layer_group) else None
        lag = self.lag_group and self.parent_flushes and local and group is not None
        lag_names: List[str] = []
        if self.parent_flushes:       # gradients are reported ready by the parent, in its order, after its joins
            P = ParamSource(P.w, P.v, P.g, ready=lag_names.append, flush=lambda: None)
        fork = self._always_fork or (group is not None and not local and not self.flush_on_exit and M >= 4096)
        side = self.branch.run if fork else (lambda fn: fn())

        def wgrad(dz_, x_, name, tag, bias=None):
            """bias = (workspace tag, bias parameter name): with the grouped launch the bias gradient's partial sums come
            out of the weight-gradient GEMM itself; otherwise a column-sum launch"""
            fused_bias = group is not None and later is not None and bias is not None and dz_.shape[0] > 512
            if group is not None:
                group.append((dz_, x_, P.g(p + name), tag, P.g(p + bias[1]) if fused_bias else None, bias))
            else:
                _wgrad(self.buf, dz_, x_, P.g(p + name), accumulate, ws_tag=tag, defer=defer)
            P.ready(p + name)
            if bias is not None:
                if fused_bias:
                    P.ready(p + bias[1])
                else:
                    dbias(bias[0], dz_, bias[1])

        def ln_bwd(which, dy, xin, mean, rstd, dxo, res):
            """LayerNorm backward; parameter gradients finished here, or their per-block partial sums left for the optimizer"""
            gw, gb = P.g(p + which + ".weight"), P.g(p + which + ".bias")
            if later is None:
                hip.layernorm_bwd(dy, xin, P.v(p + which + ".weight"), mean, rstd, dxo, gw, gb, lnws, res=res,
                                  accumulate=accumulate)
            else:
                nb = hip.layernorm_bwd_workspace_bytes(M, d)
                ws = self.buf.bytes(tg + ".lnws." + which, nb)              # one per LayerNorm: read at the end of the step
                hip.layernorm_bwd(dy, xin, P.v(p + which + ".weight"), mean, rstd, dxo, None, None, ws, res=res)
                parts = nb // (2 * d * 4)
                part = ws[:nb].view(torch.float32).view(2 * parts, d)
                later.append((part[:parts], parts, gw))
                later.append((part[parts:], parts, gb))
            P.ready(p + which + ".weight"); P.ready(p + which + ".bias")

        def dbias(tag, dz, name):
            if later is None or dz.shape[0] <= 512:
                _colsum(self.buf, tag, dz, P.g(p + name), accumulate)
            else:
                part = self.buf.get(tag + ".colsum", ((dz.shape[0] + 127) // 128, dz.shape[1]), torch.float32)
                hip.segment_colsum(dz, part, seg=128, mode=0)
                later.append((part, part.shape[0], P.g(p + name)))
            P.ready(p + name)
        # LN2: d(f2 + x1)
        ds2 = g(tg + ".ds2", (M, d), dt)
        dz1 = g(tg + ".dz1", (M, self.ffn), dt)
        dx1 = g(tg + ".dx1", (M, d), dt)
        fused_ffn = getattr(self, "_ffn_fused", False)
        ds1 = g(tg + ".ds1", (M, d), dt)
        dattn = g(tg + ".dattn", (B, T, d), dt)
        if fused_ffn:
            # LayerNorm2 backward + both dgrad GEMMs of the feed-forward sublayer + LayerNorm1 backward + the out-projection's
            # dgrad in ONE launch (csrc/ffn_chain.hip); `f2` / `a` are the saved LayerNorm inputs.  dgamma / dbeta leave as
            # per-panel partial sums, finished with the layer's other partials.
            if later is None:
                raise hip.HipError("fused feed-forward backward needs the deferred partial-sum path (bf16, M >= 4096)")
            Ta = getattr(self, "_att_T", 0)
            nwg = hip.ffn_chain_workgroups(M, d, self.ffn, Ta)
            part = self.buf.get(tg + ".ffnpart", (4 * nwg, d), torch.float32)
            att_bwd = None
            if Ta:
                # the whole layer in this launch: attention backward + in-projection dgrad + residual addend behind the
                # out-projection's dgrad (dattn never leaves the workgroup)
                if qkv_head is not None:
                    raise hip.HipError("TransformerLayerPlan.backward: a QKV head and the attention tail exclude each other")
                dqkv = g(tg + ".dqkv", (B, T, 3 * d), dt)
                dx_att = g(tg + ".dx", (B, T, d), dt)
                att_bwd = (qkv.view(M, 3 * d), lse, dqkv.view(M, 3 * d), dx_att.view(M, d), Ta)
            hip.ffn_chain_bwd(None if qkv_head is not None else dx2.view(M, d), f2, m2, r2, P.v(p + "norm2.weight"),
                              self.buf.get(tg + ".ffnpk", (hip.ffn_chain_packed_elems(d, self.ffn),), dt),
                              self.buf.get(tg + ".ffnmask", (hip.ffn_chain_mask_bytes(M, d, self.ffn, Ta),), torch.uint8),
                              ds2, dz1, None, part,
                              attn_out=(a, m1, r1, P.v(p + "norm1.weight"), ds1, None if Ta else dattn.view(M, d)),
                              qkv_head=qkv_head, attn_bwd=att_bwd)
            later.append((part[:nwg], nwg, P.g(p + "norm2.weight")))
            later.append((part[nwg:2 * nwg], nwg, P.g(p + "norm2.bias")))
            P.ready(p + "norm2.weight"); P.ready(p + "norm2.bias")
        else:
            if qkv_head is not None:
                raise hip.HipError("TransformerLayerPlan.backward: a QKV head needs the fused token-local launch")
            ln_bwd("norm2", dx2.view(M, d), f2, m2, r2, ds2, x1)
        # ds2 = d(x1 + Drop(f2)): the residual path takes it as is, the feedforward path through dropout2's mask
        df2 = ds2
        if drop:
            df2 = g(tg + ".df2", (M, d), dt)
            hip.dropout(ds2, df2, drop[0], self.seed + 2, drop[2], drop[3])

        def g_ffn2():
            wgrad(df2, f1, "feedforward.2.weight", tg + ".ws2", bias=(tg + ".b2", "feedforward.2.bias"))
        side(g_ffn2)
        if not fused_ffn:
            self._dgrad(P, df2, "feedforward.2.weight", dz1, act_below="relu", aux=f1)

        def g_ffn1():
            wgrad(dz1, x1, "feedforward.0.weight", tg + ".ws1", bias=(tg + ".b1", "feedforward.0.bias"))
        side(g_ffn1)
        if fused_ffn:
            later.append((part[2 * nwg:3 * nwg], nwg, P.g(p + "norm1.weight")))
            later.append((part[3 * nwg:], nwg, P.g(p + "norm1.bias")))
            P.ready(p + "norm1.weight"); P.ready(p + "norm1.bias")
        else:
            self._dgrad(P, dz1, "feedforward.0.weight", dx1, addend=ds2)       # + residual path
            # LN1: d(a + x)
            ln_bwd("norm1", dx1, a, m1, r1, ds1, x)
        da = ds1                       # d(x + Drop(a)): dropout1's mask on the attention path only
        if drop:
            da = g(tg + ".da", (M, d), dt)
            hip.dropout(ds1, da, drop[0], self.seed + 1, drop[2], drop[3])

        def g_out():
            wgrad(da, attn.view(M, d), "multihead_attention.out_proj.weight", tg + ".wso",
                  bias=(tg + ".bo", "multihead_attention.out_proj.bias"))
        side(g_out)
        att_done = fused_ffn and bool(getattr(self, "_att_T", 0))
        if self.split_tail and group and fused_ffn and not lag and not TU.no_tail_split and not att_done:
            # the layer whose backward runs last has nothing behind it to hide its weight-gradient launch: the three
            # problems whose operands the fused launch has just written go off NOW, beside the attention backward and the
            # in-projection's dgrad (measured and dropped for the other layers, round 4: 5 + 10 slabs per weight instead
            # of 4 and one more fork per layer against launches that already fill the chip: 1.97 -> 2.19 ms per step)
            early = list(group)
            del group[:]
            side = self.branch.run            # the rest of the group follows on the same stream (it may read these slabs)

            def run_early():
                for pr in _wgrad_group(self.buf, early, defer, later):
                    btag, bname = pr[5]
                    bp = self.buf.get(btag + ".colsum", ((pr[0].shape[0] + 127) // 128, pr[0].shape[1]), torch.float32)
                    hip.segment_colsum(pr[0], bp, seg=128, mode=0)
                    later.append((bp, bp.shape[0], P.g(p + bname)))
            self.branch.run(run_early)
        if not fused_ffn:
            self._dgrad(P, da, "multihead_attention.out_proj.weight", dattn.view(M, d))
        dqkv = g(tg + ".dqkv", (B, T, 3 * d), dt)
        if not att_done:
            hip.attention_bwd(qkv, attn, dattn, lse, dqkv, self.h, drop=drop)
        dq2 = dqkv.view(M, 3 * d)

        def g_in():
            wgrad(dq2, x, "multihead_attention.in_proj_weight", tg + ".wsi",
                  bias=(tg + ".bi", "multihead_attention.in_proj_bias"))
        side(g_in)
        if att_done:
            dx = g(tg + ".dx", (B, T, d), dt)      # written by the fused launch
        elif self.head_active(M):
            dx = None                  # dq2 . Wqkv + ds1 is computed in front of the layer below's fused backward launch
            self.pending_head = (self.packed_image(), dq2, ds1)
        else:
            dx = g(tg + ".dx", (B, T, d), dt)
            self._dgrad(P, dq2, "multihead_attention.in_proj_weight", dx.view(M, d), addend=ds1)
        if group:
            def run_group():
                # problems the grouped launch could not take sum their bias gradient the plain way (P.ready already said)
                for pr in _wgrad_group(self.buf, group, defer, later):
                    btag, bname = pr[5]
                    part = self.buf.get(btag + ".colsum", ((pr[0].shape[0] + 127) // 128, pr[0].shape[1]), torch.float32)
                    hip.segment_colsum(pr[0], part, seg=128, mode=0)
                    later.append((part, part.shape[0], P.g(p + bname)))
                if local:     # reads the slabs of the launch above: same stream
                    hip.step_reduce_parts(defer, [(part, rows, dst) for part, rows, dst in later])
            if lag:
                self._lagged = (run_group, lag_names)
                return dx
            side(run_group)
        elif local:
            hip.step_reduce_parts(defer, [(part, rows, dst) for part, rows, dst in later])
        if self.join_on_exit or self.flush_on_exit:
            self.branch.join()
        if self.parent_flushes:
            self._lagged = (None, lag_names)          # everything issued; the parent reports and flushes
        elif self.flush_on_exit:
            P.flush()
        return dx


# ------------------------------------------------------------------------------------------------
class TimeMLPPlan:
    """e = W2 silu(W1 sinus(t) + b1) + b2; sinus(t) is a row gather from a host-built fp32 table."""

    def __init__(self, dtype, buf: Buffers, tag="tm"):
        self.dtype, self.buf, self.tag = dtype, buf, tag
        self.ctx = None

    def fused_ok(self, table: torch.Tensor, P: ParamSource) -> bool:
        w1, w2 = P.w("time_mlp.0.weight"), P.w("time_mlp.2.weight")
        return self.dtype == torch.bfloat16 and not TU.no_time_fuse \
            and hip.time_mlp_fwd_supported(table.shape[1], w1.shape[0], w2.shape[0])

    def forward(self, t: torch.Tensor, table: torch.Tensor, P: ParamSource, pack=None, slots=None) -> torch.Tensor:
        """pack: (weights, packed, D, H) of the chain kernel -- packed by the same launch when the fused kernel runs
        (the caller checks fused_ok() first)"""
        B = t.shape[0]
        g, dt, tg = self.buf.get, self.dtype, self.tag
        w1, w2 = P.w("time_mlp.0.weight"), P.w("time_mlp.2.weight")
        s = g(tg + ".s", (B, table.shape[1]), dt)
        u = g(tg + ".u", (B, w1.shape[0]), dt)
        zu = g(tg + ".zu", (B, w1.shape[0]), dt)
        e = g(tg + ".e", (B, w2.shape[0]), dt)
        if self.fused_ok(table, P):
            # one launch instead of gather + two M = B GEMMs (three latency-bound launches on the critical path)
            hip.time_mlp_fwd(table, t, w1, P.v("time_mlp.0.bias"), w2, P.v("time_mlp.2.bias"), s, zu, u, e, pack=pack,
                             slots=slots if pack is not None else None)
            self.ctx = (s, u, zu)
            return e
        hip.gather_rows(table, t, s)
        hip.linear_fwd(s, w1, P.v("time_mlp.0.bias"), u, act="silu", z=zu)
        hip.linear_fwd(u, w2, P.v("time_mlp.2.bias"), e)
        self.ctx = (s, u, zu)
        return e

    @staticmethod
    def ready_order() -> List[str]:
        return ["time_mlp.2.weight", "time_mlp.2.bias", "time_mlp.0.weight", "time_mlp.0.bias"]

    def _de(self, de32: torch.Tensor, de_lp: Optional[torch.Tensor]) -> torch.Tensor:
        if self.dtype == torch.float32:
            return de32
        return de_lp if de_lp is not None else _as_dtype(self.buf, self.tag + ".de", de32, self.dtype)

    def backward_out_layer(self, de32, de_lp, P: ParamSource, accumulate=False, defer=None, ready=True):
        """gradients of the output layer (time_mlp.2): independent of the dgrad chain below.
        de32 None: the caller produces the bias gradient itself (chain path: one multi-segment reduction)."""
        s, u, zu = self.ctx
        tg = self.tag
        _wgrad(self.buf, self._de(de32, de_lp), u, P.g("time_mlp.2.weight"), accumulate, ws_tag=tg + ".ws2", defer=defer)
        if ready:
            P.ready("time_mlp.2.weight")
        if de32 is not None:
            _colsum(self.buf, tg + ".b2", de32, P.g("time_mlp.2.bias"), accumulate)
            if ready:
                P.ready("time_mlp.2.bias")

    def fused_bwd_ok(self, P: ParamSource, defer, accumulate) -> bool:
        """the one-launch backward of the hidden layer (csrc/chain.hip: ib_time_mlp_bwd) leaves fp32 partial slabs, so it
        needs a deferred reduction to hand them to (the step's optimizer / ib_step_reduce)"""
        if self.dtype != torch.bfloat16 or defer is None or accumulate or TU.no_time_bwd_fuse:
            return False
        w1, w2 = P.w("time_mlp.0.weight"), P.w("time_mlp.2.weight")
        return hip.time_mlp_bwd_supported(w1.shape[1], w1.shape[0], w2.shape[0])

    def backward_hidden(self, de32, de_lp, P: ParamSource, accumulate=False, defer=None, ready=True, rider=None):
        """rider (a list): with the one-launch backward available, do NOT launch it -- append its operands instead (the
        caller puts its workgroups into the grouped weight-gradient launch, or launches ib_time_mlp_bwd itself)"""
        s, u, zu = self.ctx
        tg = self.tag
        if self.fused_bwd_ok(P, defer, accumulate):
            de = self._de(de32, de_lp)
            hid, temb = P.w("time_mlp.0.weight").shape
            nsl = hip.time_mlp_bwd_slab_count(de.shape[0])
            wsw = self.buf.get(tg + ".bww", (nsl, hid, temb), torch.float32)
            wsb = self.buf.get(tg + ".bwb", (nsl, hid), torch.float32)
            if rider is not None:                    # launched by the caller inside its grouped weight-gradient launch
                rider.append((de, P.w("time_mlp.2.weight"), zu, s, wsw, wsb))
            else:
                hip.time_mlp_bwd(de, P.w("time_mlp.2.weight"), zu, s, wsw, wsb)
            defer.append((wsw, nsl, P.g("time_mlp.0.weight")))
            defer.append((wsb, nsl, P.g("time_mlp.0.bias").view(1, hid)))
            if ready:
                P.ready("time_mlp.0.weight"); P.ready("time_mlp.0.bias")
            return
        du = self.buf.get(tg + ".du", u.shape, self.dtype)
        de = self._de(de32, de_lp)
        # one row per window: the few-row kernel (16 output columns per workgroup, the bias sums in the same launch);
        # shapes it does not take go through the tiled dgrad + a column-sum launch
        fused = self.dtype == torch.bfloat16 and not TU.no_skinny and \
            hip.linear_dgrad_skinny(de, P.w("time_mlp.2.weight"), du, act_below="silu", aux=zu,
                                    dbias=P.g("time_mlp.0.bias"), accumulate=accumulate)
        if not fused:
            hip.linear_dgrad(de, P.w("time_mlp.2.weight"), du, act_below="silu", aux=zu)
        _wgrad(self.buf, du, s, P.g("time_mlp.0.weight"), accumulate, ws_tag=tg + ".ws0", defer=defer)
        if ready:
            P.ready("time_mlp.0.weight")
        if not fused:
            _colsum(self.buf, tg + ".b1", du, P.g("time_mlp.0.bias"), accumulate)
        if ready:
            P.ready("time_mlp.0.bias")

    def backward(self, de32: torch.Tensor, P: ParamSource, accumulate=False, de_lp: Optional[torch.Tensor] = None):
        """both halves on the current stream (callers that want them concurrent fork the halves as SIBLING
        branches of the main stream: a fork nested inside a forked stream crashes hipStreamEndCapture)"""
        self.backward_out_layer(de32, de_lp, P, accumulate)
        self.backward_hidden(de32, de_lp, P, accumulate)


class DenoiserMLPPlan:
    """Token-wise MLP denoiser (BASELINE config 2): per block h = LN(silu(W h + b + e[window])).

    Backward = a critical chain (head dgrad -> LN bwd -> dgrad -> LN bwd -> time-MLP bwd) plus work that only
    hangs off it (the weight-gradient GEMMs and bias sums).  The hangers run on forked streams (`Branch`), so
    these individually latency-bound launches overlap; under hipGraph capture they become parallel graph
    branches.  Every concurrent wgrad owns its slab workspace."""

    def __init__(self, hidden: Sequence[int], dtype, device):
        self.hidden, self.dtype = list(hidden), dtype
        self.buf = Buffers(device)
        self.time = TimeMLPPlan(dtype, self.buf)
        self.ctx = None
        # the trainer switches the branches off under data parallelism (gradient-bucket events are recorded on
        # ONE stream)
        self.br_head = Branch(device, name="head")
        self.br_blk = [Branch(device, name="blk") for _ in self.hidden]
        self.branch = Branch(device, name="time_bwd")   # time-MLP backward: hidden-layer chain
        self.br_tout = Branch(device, name="time_out")  # time-MLP backward: output-layer gradients
        self.br_tfwd = Branch(device, name="time_fwd")  # time-MLP forward (beside q_sample + the first block's GEMM)
        self.br_pack = Branch(device, name="pack")      # chain path: weight packing beside the time-MLP forward
        self.fuse_reduce_into_optimizer = False         # set by HipTrainer for single-GPU steps
        self.slots_used = False                         # the last chain_step read its batch through input slots
        self.pending_sources = None

    def branches(self) -> List[Branch]:
        return [self.br_head, self.branch, self.br_tout, self.br_tfwd, self.br_pack] + self.br_blk

    def forward(self, x, t: torch.Tensor, table: torch.Tensor, P: ParamSource,
                out: Optional[torch.Tensor] = None, BT: Optional[Tuple[int, int]] = None) -> torch.Tensor:
        """x: [B,T,D] contiguous, or a (possibly row-padded) 2-D [B*T, D] view with BT=(B,T)"""
        if x.dim() == 3:
            B, T, D = x.shape
            h = x.view(B * T, D)
        else:
            B, T = BT
            D = x.shape[1]
            h = x
        M = B * T
        g, dt = self.buf.get, self.dtype
        # the time embedding is added in the LayerNorm prologue (not the GEMM epilogue), so the time-MLP runs on a
        # forked stream beside the first GEMM; it is joined right before the first LayerNorm
        eref = {}
        self.br_tfwd.run(lambda: eref.__setitem__("e", self.time.forward(t, table, P)))
        saved = []
        off = 0
        for i, hd in enumerate(self.hidden):
            z = g(f"dm.z{i}", (M, hd), dt)
            hip.linear_fwd(h, P.w(f"blocks.{i}.linear.weight"), P.v(f"blocks.{i}.linear.bias"), z)
            if i == 0:
                self.br_tfwd.join()
            e = eref["e"]
            hn = g(f"dm.hn{i}", (M, hd), dt)
            mu, rs = g(f"dm.mu{i}", (M,), torch.float32), g(f"dm.rs{i}", (M,), torch.float32)
            hip.layernorm_fwd(z, P.v(f"blocks.{i}.norm.weight"), P.v(f"blocks.{i}.norm.bias"), hn, mu, rs, act="silu",
                              add_div=e[:, off:off + hd], seg=T)
            saved.append((h, z, mu, rs))
            h = hn
            off += hd
        e = eref["e"]
        if out is None:
            out = g("dm.out", (B, T, D), dt)
        out2 = out.view(M, D) if out.dim() == 3 else out
        hip.linear_fwd(h, P.w("head.weight"), P.v("head.bias"), out2)
        self.ctx = (saved, h, B, T, e)
        return out

    def ready_order(self) -> List[str]:
        o = ["head.weight", "head.bias"]
        for i in range(len(self.hidden) - 1, 0, -1):
            o += [f"blocks.{i}.norm.weight", f"blocks.{i}.norm.bias", f"blocks.{i}.linear.weight",
                  f"blocks.{i}.linear.bias"]
        return o + TimeMLPPlan.ready_order() + ["blocks.0.norm.weight", "blocks.0.norm.bias",
                                                "blocks.0.linear.weight", "blocks.0.linear.bias"]

    def backward(self, dout, P: ParamSource, accumulate=False):
        saved, hlast, B, T, e = self.ctx
        M = B * T
        g, dt = self.buf.get, self.dtype
        dout = dout.view(M, -1) if dout.dim() == 3 else dout

        def head_grads():
            _wgrad(self.buf, dout, hlast, P.g("head.weight"), accumulate, ws_tag="dm.wsH")
            P.ready("head.weight")
            _colsum(self.buf, "dm.bh", dout, P.g("head.bias"), accumulate)
            P.ready("head.bias")
        self.br_head.run(head_grads)
        dh = g("dm.dh_last", hlast.shape, dt)
        hip.linear_dgrad(dout, P.w("head.weight"), dh)
        H = sum(self.hidden)
        de32 = g("dm.de32", (B, H), torch.float32)
        de_lp = g("dm.de_lp", (B, H), torch.bfloat16) if dt == torch.bfloat16 else None
        off = H
        for i in range(len(self.hidden) - 1, -1, -1):
            hd = self.hidden[i]
            off -= hd
            hin, z, mu, rs = saved[i]
            # the LayerNorm parameter-gradient reduction is deferred to the block's side work (own workspace)
            lnws = self.buf.bytes(f"dm.lnws{i}", hip.layernorm_bwd_workspace_bytes(M, hd))
            dz = g(f"dm.dz{i}", (M, hd), dt)
            hip.layernorm_bwd(dh, z, P.v(f"blocks.{i}.norm.weight"), mu, rs, dz, None, None, lnws, act="silu",
                              add_div=e[:, off:off + hd], seg=T)
            sl = de32[:, off:off + hd]

            def blk_grads(i=i, dz=dz, hin=hin, sl=sl, lnws=lnws, hd=hd):
                hip.layernorm_bwd_reduce(lnws, P.g(f"blocks.{i}.norm.weight"), P.g(f"blocks.{i}.norm.bias"), M, hd,
                                         accumulate=accumulate)
                P.ready(f"blocks.{i}.norm.weight"); P.ready(f"blocks.{i}.norm.bias")
                _wgrad(self.buf, dz, hin, P.g(f"blocks.{i}.linear.weight"), accumulate, ws_tag=f"dm.ws{i}")
                P.ready(f"blocks.{i}.linear.weight")
                _colsum(self.buf, f"dm.b{i}", sl, P.g(f"blocks.{i}.linear.bias"), accumulate)
                P.ready(f"blocks.{i}.linear.bias")

            # per-window sums of dz: the time-embedding gradient AND (summed over windows) the bias gradient;
            # the bf16 copy is the time-MLP backward's GEMM operand
            hip.segment_colsum(dz, sl, seg=T, mode=0, out_bf16=None if de_lp is None else de_lp[:, off:off + hd])
            if i > 0:
                self.br_blk[i].run(blk_grads)
                dh = g(f"dm.dh{i - 1}", hin.shape, dt)
                hip.linear_dgrad(dz, P.w(f"blocks.{i}.linear.weight"), dh)
            else:
                # de32 is complete: the two halves of the time-MLP backward fork off as siblings; block 0's own
                # gradients stay on the main stream
                self.br_tout.run(lambda: self.time.backward_out_layer(de32, de_lp, P, accumulate))
                self.branch.run(lambda: self.time.backward_hidden(de32, de_lp, P, accumulate))
                blk_grads()
        for b in self.branches():
            b.join()


    # ---- fused chain: q_sample + forward + loss + dgrad chain in ONE launch (csrc/chain.hip) ------------------
    def chain_ok(self, D: int) -> bool:
        if TU.no_chain or self.dtype != torch.bfloat16 or not self.hidden:
            return False
        H = self.hidden[0]
        return all(h == H for h in self.hidden) and hip.mlp_chain_supported(D, H, len(self.hidden))

    def chain_step(self, x0: torch.Tensor, eps: torch.Tensor, t: torch.Tensor, tabs, P: ParamSource,
                   result: torch.Tensor, accumulate=False, slots: Optional[torch.Tensor] = None):
        """the whole diffusion training step up to the gradients (HipTrainer's diffusion path for this model):
        [weight pack || time-MLP forward] -> chain kernel -> {weight-gradient GEMMs (slabs only), one multi-segment
        reduction for every small gradient + the loss, time-MLP backward} -> one slab reduction for all GEMMs."""
        B, T, D = x0.shape
        M, L, H = B * T, len(self.hidden), self.hidden[0]
        g, dt = self.buf.get, self.dtype
        names = [f"blocks.{i}.linear.weight" for i in range(L)] + ["head.weight"]
        packed = g("ch.packed", (hip.mlp_chain_packed_elems(D, H, L),), dt)
        # the weight packing rides in the time-MLP forward's launch (every fork / join of the captured graph costs
        # tens of microseconds here: they are independent, but NOT worth a branch)
        weights = [P.w(n) for n in names]
        self.slots_used = False
        if self.time.fused_ok(tabs.temb, P):
            # `slots` (device array {x0, eps, t}): both launches read the batch through it, so a captured graph can
            # consume every step's batch where it lies (HipTrainer sets the slots before each replay)
            e = self.time.forward(t, tabs.temb, P, pack=(weights, packed, D, H), slots=slots)   # [B, L*H]
            self.slots_used = slots is not None
        else:
            hip.mlp_chain_pack(weights, packed, D, H)
            e = self.time.forward(t, tabs.temb, P)
        Dp = (D + 7) // 8 * 8
        xt = g("ch.xt", (M, Dp), dt)[:, :D]
        dpred = g("ch.dpred", (M, Dp), dt)[:, :D]
        # the pre-activations stay in the chain kernel's registers (L <= 2); deeper stacks pass them through HBM
        u = None if (L <= 2 and not TU.chain_v1) else [g(f"ch.u{i}", (M, H), dt) for i in range(L)]
        h = [g(f"ch.h{i}", (M, H), dt) for i in range(L)]
        dz = [g(f"ch.dz{i}", (M, H), dt) for i in range(L)]
        nwg = hip.mlp_chain_workgroups(M)
        W = hip.mlp_chain_partial_width(D, H, L)
        part = g("ch.part", (nwg, W), torch.float32)
        Hs = L * H
        de_lp = g("dm.de_lp", (B, Hs), torch.bfloat16)
        window_panels = hip.mlp_chain_rows_per_workgroup(M) == T and nwg == B   # panel == window: de comes for free
        hip.mlp_chain_train(x0, eps, t, tabs.sqrt_ab, tabs.sqrt_1mab, e, packed,
                            [P.v(f"blocks.{i}.linear.bias") for i in range(L)] + [P.v("head.bias")],
                            [P.v(f"blocks.{i}.norm.weight") for i in range(L)],
                            [P.v(f"blocks.{i}.norm.bias") for i in range(L)], xt, u, h, dz, dpred, part, T,
                            de_lp=de_lp if window_panels else None, slots=slots if self.slots_used else None)

        # every gradient operand now sits in HBM.  Issue order = the order the graph's ready nodes get the machine:
        # the time-MLP backward first (a dependent chain of small launches; started late it becomes the step's tail),
        # then the large weight-gradient GEMMs, one branch each; the main stream does the small reductions.
        defer = None if TU.no_defer else []
        de32 = None
        if not window_panels:
            de32 = g("dm.de32", (B, Hs), torch.float32)
            for i in range(L):
                hip.segment_colsum(dz[i], de32[:, i * H:(i + 1) * H], seg=T, mode=0, out_bf16=de_lp[:, i * H:(i + 1) * H])
        # Streams: a fifth concurrent branch of the captured graph only started when another finished, and every fork /
        # join costs tens of microseconds (same box: main + 2 branches 0.282 ms, + a head branch 0.308, none 0.320).
        # So: ONE branch for the dependent time-MLP chain; every independent weight-gradient GEMM of the step
        # (head, blocks, time_mlp.2) goes into ONE grouped launch on the main stream.
        # (issued AFTER the grouped launch instead, the branch's first kernel only started when the grouped launch had
        # finished -- no overlap at all: 0.232 -> 0.250 ms/step)
        rider_ops = []
        if TU.skip_time_bwd and hip.measurement_build():
            # TIMING-ONLY (wrong gradients): an upper bound of what the time-MLP backward costs.  Honoured only while the
            # measurement build of the library is loaded (tools/, IB_HIP_LIB) -- the product ignores the variable
            if defer is not None and getattr(self, "_tb_defer", None):
                defer.extend(self._tb_defer)
        else:
            n0 = len(defer) if defer is not None else 0
            if self.time.fused_bwd_ok(P, defer, accumulate):
                # one set of independent workgroups, no fork / join of the captured graph: they ride in the grouped
                # weight-gradient launch below (204 work items for 256 CUs: the launch has 52 CUs to spare), or run as a
                # launch of their own on the main stream when that launch does not take them
                rider_ops = []
                self.time.backward_hidden(de32, de_lp, P, accumulate, defer=defer, ready=False,
                                          rider=None if TU.no_tb_rider else rider_ops)
            else:
                self.branch.run(lambda: self.time.backward_hidden(de32, de_lp, P, accumulate, defer=defer, ready=False))
            if defer is not None:
                self._tb_defer = defer[n0:]
        grouped = defer is not None and not accumulate
        probs = [(dpred, h[L - 1], P.g("head.weight"), "dm.wsH")]
        for i in range(L - 1, -1, -1):
            probs.append((dz[i], h[i - 1] if i > 0 else xt, P.g(f"blocks.{i}.linear.weight"), f"dm.ws{i}"))
        s_, u_, zu_ = self.time.ctx
        probs.append((de_lp, u_, P.g("time_mlp.2.weight"), self.time.tag + ".ws2"))     # B rows only: rides along
        tb = [rider_ops[0], False] if rider_ops else None
        if grouped and len(probs) <= 6:
            _wgrad_group(self.buf, probs, defer, time_bwd=tb)
        else:
            for dz_, x_, dw_, tag in probs:
                _wgrad(self.buf, dz_, x_, dw_, accumulate, ws_tag=tag, defer=defer)
        if tb is not None and not tb[1]:
            hip.time_mlp_bwd(*tb[0])
        # main stream: every small gradient (LayerNorm gains / biases, linear biases, head bias, time_mlp.2.bias) and
        # the loss in one launch
        tb2 = P.g("time_mlp.2.bias")
        segs = []
        for i in range(L):
            segs += [(3 * i * H, H, P.g(f"blocks.{i}.norm.weight"), None, 1.0),
                     (3 * i * H + H, H, P.g(f"blocks.{i}.norm.bias"), None, 1.0),
                     (3 * i * H + 2 * H, H, P.g(f"blocks.{i}.linear.bias"), tb2[i * H:(i + 1) * H], 1.0)]
        segs += [(3 * L * H, D, P.g("head.bias"), None, 1.0), (W - 4, 1, result, None, 1.0 / (M * D))]
        merged = bool(defer) and not accumulate and len(defer) <= 8
        if not merged:
            if accumulate:     # the loss scalar is never accumulated
                hip.colsum_segments(part, nwg, segs[:-1], accumulate=True)
                hip.colsum_segments(part, nwg, segs[-1:], accumulate=False)
            else:
                hip.colsum_segments(part, nwg, segs, accumulate=False)
        for b in self.branches():
            b.join()
        self.pending_sources = None
        if merged and self.fuse_reduce_into_optimizer:
            # single GPU: the optimizer sums the slabs / partial rows itself (ib_optim_step_sources)
            self.pending_sources = (list(defer), part, nwg, segs)
        elif merged:             # every slab set + every small gradient + the loss: one launch
            hip.step_reduce(defer, part, nwg, segs)
        elif defer:
            hip.slab_reduce_multi(defer, accumulate=accumulate)
        for n in self.ready_order():
            P.ready(n)


class DenoiserTransformerPlan:
    """Transformer denoiser (BASELINE configs 3-5): in-proj(x ++ frame-embedding) + time embedding,
    N reference TransformerLayers, out-proj."""

    def __init__(self, feat: int, pos_dim: int, d_model: int, num_heads: int, ffn: int, num_layers: int, dtype, device):
        self.D, self.Pd, self.d, self.dtype = feat, pos_dim, d_model, dtype
        self.buf = Buffers(device)
        self.time = TimeMLPPlan(dtype, self.buf)
        self.layers = [TransformerLayerPlan(f"transformer_layers.{l}.", d_model, num_heads, ffn, dtype, device,
                                            buf=self.buf, tag=f"tl{l}") for l in range(num_layers)]
        for lp in self.layers:
            lp.join_on_exit = False          # joined once, at the end of the whole backward
            lp.own_wt = False                # the transposed weight copies of ALL layers are refreshed by one launch
        for lo, hi in zip(self.layers[:-1], self.layers[1:]):
            lo.qkv_tail_for, hi.qkv_dgrad_below = hi, lo     # the upper layer's in-projection rides in the lower layer's launches
        self.layers[0].split_tail = True
        self.ctx = None
        self._posproj_T = None
        self.fuse_reduce_into_optimizer = False         # set by HipTrainer for single-GPU steps
        self.pending_sources = None
        self.early_optimizer = None                     # set by HipTrainer: fn(parameter prefix, sources) -- see backward()
        # the step's head and tail are chains of SMALL launches (time-MLP, frame-embedding projection, weight transposes;
        # their gradients): latency-bound and mutually independent, so they run as sibling branches of the main stream
        # (parallel branches of the captured graph).  The trainer switches them off under data parallelism.
        self.br_time = Branch(device, name="tr_time")
        self.br_thid = Branch(device, name="tr_thid")
        self.br_pos = Branch(device, name="tr_pos")
        self.br_wt = Branch(device, name="tr_wt")
        self.br_side = Branch(device, name="tr_side")    # sampler: the windows beyond the fused launch's last full round

    def side_windows(self, B: int, T: int) -> int:
        """sampler: how many of the batch's last windows take the per-op side stack (0: none) -- the fused launch's last
        round of panels would fill at most `infer_split_max_rem` of the 256 CUs"""
        M = B * T
        if (not self.inference or TU.no_infer_split or TU.no_infer_chain or not self.br_side.on or M < TU.infer_chain_min_m
                or self.dtype != torch.bfloat16 or self.d != 512 or not all(lp.infer_packed for lp in self.layers)):
            return 0
        panels = -(-M // 64)
        rounds, rem = divmod(panels, 256)
        if rounds == 0 or rem == 0 or rem > TU.infer_split_max_rem:
            return 0
        Bm = (rounds * 256 * 64) // T
        Bs = B - Bm
        if Bs <= 0 or Bm * T < TU.infer_chain_min_m or Bs * T > min(TU.linln_panel_max_m, TU.ffn_infer_max_m):
            return 0
        return Bs

    def side_layers(self) -> List["TransformerLayerPlan"]:
        """a second set of layer plans over the same parameters and packed images, with activation buffers of their own"""
        if self._side is None:
            self._side = []
            for lp in self.layers:
                sl = TransformerLayerPlan(lp.p, lp.d, lp.h, lp.ffn, lp.dtype, self.buf.device, buf=self.buf, tag=lp.tag + "s")
                sl.join_on_exit, sl.own_wt = False, False
                sl.packed_image = lp.packed_image             # the main layer's image (packed once per sampling loop)
                self._side.append(sl)
        for sl, lp in zip(self._side, self.layers):
            sl.inference, sl.infer_packed = lp.inference, lp.infer_packed
        return self._side

    _side = None

    def flush_each_layer(self, on: bool):
        """overlapped data-parallel steps: keep the layers' side streams, hand completed gradient buckets to the trainer at
        every layer boundary (ParamSource.flush) instead of running the whole backward on one stream"""
        lag = bool(on) and not TU.no_lag_group
        for i, lp in enumerate(self.layers):
            lp.flush_on_exit = bool(on)
            lp.parent_flushes = lag
            lp.lag_group = lag                      # (round 5: layer 0's too -- it runs beside the step's tail of small launches)

    def set_inference(self, on: bool):
        """forward-only mode with frozen weights (the DDIM sampler): fused Linear + residual + LayerNorm in every layer,
        the frame-embedding projection computed once"""
        self.inference = bool(on)
        self._posproj_T = None
        self._e_all = None
        for lp in self.layers:
            lp.inference = bool(on)
            lp.infer_packed = False                       # prepare_inference packs the (then frozen) weights again

    def prepare_inference(self, P: ParamSource, T: int, D: int, table: Optional[torch.Tensor] = None):
        """once per sampling loop (weights frozen from here on): the frame-embedding half of the input projection, and --
        given the sinusoid table -- the time embedding of EVERY timestep (one time-MLP forward over all table rows), so a
        denoise step only gathers its rows instead of running the time-MLP (15 us of a 0.5 ms step at B = 16)"""
        w_in = P.w("in_proj.weight")
        pos = P.w("temporal_embedding.embedding.weight")[:T]
        posproj = self.buf.get("dt.posproj", (T, self.d), self.dtype)
        hip.tiny_matmul(pos, w_in[:, D:].t(), posproj)
        self._posproj_T = T
        self._e_all = None
        # zero-padded copies of the two D-wide projections (D = 300: rows of 600 bytes are not 16-byte aligned, so both ran on
        # the generic register-staged kernel: 18.6 + 13.5 us of a 380-us step at B = 16).  With the sampler's state and
        # noise buffers pitched to infer_pitch(D) columns (pad columns zero) the input projection reduces over 320 columns on
        # the LDS-DMA ring kernel and the output projection writes 320 columns on the 256 x 128 NT kernel.
        self._pad = None
        Kp = self.infer_pitch(D)
        if Kp != D and self.dtype == torch.bfloat16:
            w_in_pad = self.buf.get("dt.w_in_pad", (self.d, Kp), self.dtype)
            w_out_pad = self.buf.get("dt.w_out_pad", (Kp, self.d), self.dtype)
            b_out_pad = self.buf.get("dt.b_out_pad", (Kp,), torch.float32)
            w_in_pad.zero_(); w_out_pad.zero_(); b_out_pad.zero_()
            hip.cast2d(w_in[:, :D], w_in_pad[:, :D])
            hip.cast2d(P.w("out_proj.weight"), w_out_pad[:D])
            b_out_pad[:D].copy_(P.v("out_proj.bias"))
            self._pad = (Kp, w_in_pad, w_out_pad, b_out_pad)
        # frozen weights: every layer's packed image once per sampling loop -- the attention out-projection + residual +
        # LayerNorm1 of a denoise step is then ONE launch over row panels (csrc/linln_panel.hip) while the step has at most a
        # few thousand rows, instead of a split-K GEMM into fp32 slabs + a reduction launch
        for lp in self.layers:
            lp.infer_packed = False
        if (self.dtype == torch.bfloat16 and self.d == 512 and not TU.no_linln_panel
                and all(hip.ffn_chain_supported(self.d, lp.ffn) for lp in self.layers)):
            hip.ffn_chain_pack([(P.w(lp.p + "feedforward.0.weight"), P.w(lp.p + "feedforward.2.weight"), lp.packed_image(),
                                 P.w(lp.p + "multihead_attention.out_proj.weight"),
                                 P.w(lp.p + "multihead_attention.in_proj_weight")) for lp in self.layers])
            for lp in self.layers:
                lp.infer_packed = True
        if table is not None and not TU.no_time_table:
            steps = table.shape[0]
            every_t = torch.arange(steps, dtype=torch.int64, device=table.device)
            e = self.time.forward(every_t, table, P)                                     # [steps, d] in the compute dtype
            e_all = self.buf.get("dt.e_all", (steps, self.d), torch.float32)
            hip.cast2d(e, e_all) if e.dtype != torch.float32 else e_all.copy_(e)
            self._e_all = e_all

    _e_all = None
    _pad = None

    inference = False

    def train_pitch(self, D: int, M: int) -> int:
        """row pitch (elements) of the trainer's D-wide activation buffers (x_t, prediction, dL/dprediction): D rounded up to
        64 when the padded projections below apply (bf16, the large-M kernels), else to 8 (16-byte aligned rows)"""
        if (self.dtype == torch.bfloat16 and M >= 4096 and D % 64 != 0
                and not (TU.no_pad or TU.no_train_pad or TU.no_nt)):
            return (D + 63) // 64 * 64
        return D if TU.no_pad else (D + 7) // 8 * 8

    def _train_pad(self, M: int, D: int, x2: torch.Tensor, out: Optional[torch.Tensor]):
        """training step with pitched activation buffers (train_pitch): zero-padded copies of the D-wide projection weights,
        refreshed every step beside the weight transposes, so that the input projection reduces over 320 columns on the
        LDS-DMA ring kernel and the output projection / its input gradient run on the 256 x 128 NT kernel (the 300-wide
        operands put all three on the generic register-staged kernel: 23.6 + 17.1 + 16.8 us per step)"""
        Kp = self.train_pitch(D, M)
        if Kp % 64 != 0 or Kp == D or out is None or x2.dim() != 2 or out.dim() != 2 or x2.stride(0) != Kp \
                or out.stride(0) != Kp or x2.stride(1) != 1:
            return None
        g, dt = self.buf.get, self.dtype
        return {"Kp": Kp, "w_in": g("dt.tp.w_in", (self.d, Kp), dt, zero=True), "w_out": g("dt.tp.w_out", (Kp, self.d), dt, zero=True),
                "w_outT": g("dt.tp.w_outT", (self.d, Kp), dt, zero=True), "b_out": g("dt.tp.b_out", (Kp,), torch.float32, zero=True)}

    _tp = None

    @staticmethod
    def infer_pitch(D: int) -> int:
        """row pitch (elements) the sampler gives its state / noise buffers: D rounded up to 64 (the K step of the LDS-DMA
        kernels) unless IB_NO_PAD; the pad columns must be zero"""
        return D if (TU.no_pad or D % 64 == 0) else (D + 63) // 64 * 64

    def branches(self) -> List[Branch]:
        return [lp.branch for lp in self.layers] + [self.br_time, self.br_thid, self.br_pos, self.br_wt, self.br_side]

    def forward(self, x3: torch.Tensor, t: torch.Tensor, table: torch.Tensor, P: ParamSource,
                out: Optional[torch.Tensor] = None, BT: Optional[Tuple[int, int]] = None) -> torch.Tensor:
        """x3 / out: contiguous [B,T,D], or (with BT=(B,T)) row-padded 2-D [B*T, D] views"""
        if x3.dim() == 2:
            (B, T), D = BT, x3.shape[1]
        else:
            B, T, D = x3.shape
        M = B * T
        x2 = x3 if x3.dim() == 2 else x3.view(M, D)
        g, dt = self.buf.get, self.dtype
        w_in = P.w("in_proj.weight")                                         # [d, D + Pd]
        pos = P.w("temporal_embedding.embedding.weight")[:T]                 # [T, Pd]
        posproj = g("dt.posproj", (T, self.d), dt)
        pairs = [pr for lp in self.layers for pr in lp.wt_pairs(P, M)]
        ffn_items = [it for it in (lp.ffn_pack_item(P, M) for lp in self.layers) if it is not None]
        if self.inference and self._e_all is not None:
            e = g("dt.e_rows", (B, self.d), dt)
            hip.gather_rows(self._e_all, t, e)                               # rows of the per-timestep table
        elif self.inference:
            e = self.time.forward(t, table, P)                               # [B, d]
        else:
            # training: the time-MLP and the weight transposes (read by the backward only) beside the projection below
            tp = self._tp = self._train_pad(M, D, x2, out)
            box = []

            def t_branch():
                box.append(self.time.forward(t, table, P))
                if tp:
                    hip.cast2d(w_in[:, :D], tp["w_in"][:, :D])
            self.br_time.run(t_branch)
            e = box[0]
            if pairs or tp or ffn_items:
                def wt_branch():
                    if ffn_items:        # first: layer 0's feed-forward sublayer is the first reader
                        hip.ffn_chain_pack(ffn_items)
                    pr = list(pairs)
                    if tp:
                        w_out = P.w("out_proj.weight")
                        pr.append((w_out, tp["w_outT"][:, :D]))
                        hip.cast2d(w_out, tp["w_out"][:D])
                        hip.cast2d(P.v("out_proj.bias").view(1, D), tp["b_out"].view(1, -1)[:, :D])
                    hip.transpose_multi(pr)
                self.br_wt.run(wt_branch)
        if not (self.inference and self._posproj_T == T):    # frozen weights (sampling): projected once per sample()
            hip.tiny_matmul(pos, w_in[:, D:].t(), posproj)
        self.br_time.join()
        h0 = g("dt.h0", (B, T, self.d), dt)
        # sampler with pitched buffers (prepare_inference): both projections over the padded width
        padded = (self.inference and self._pad is not None and out is not None and x2.dim() == 2 and out.dim() == 2
                  and x2.stride(0) == self._pad[0] and out.stride(0) == self._pad[0] and x2.stride(1) == 1
                  and M >= TU.pad_min_m)   # below: B = 4 / 8 -1.6 %, B = 16 / 32 +3 %
        tp = None if self.inference else self._tp
        if padded:
            Kp, w_in_pad, w_out_pad, b_out_pad = self._pad
            hip.linear_fwd(x2.as_strided((M, Kp), (Kp, 1)), w_in_pad, P.v("in_proj.bias"), h0.view(M, self.d), add_div=e,
                           add_mod=posproj, seg=T)
        elif tp:
            hip.linear_fwd(x2.as_strided((M, tp["Kp"]), (tp["Kp"], 1)), tp["w_in"], P.v("in_proj.bias"), h0.view(M, self.d),
                           add_div=e, add_mod=posproj, seg=T)
        else:
            hip.linear_fwd(x2, w_in[:, :D], P.v("in_proj.bias"), h0.view(M, self.d), add_div=e,
                           add_mod=posproj, seg=T)
        h = h0
        if ffn_items and not self.inference:
            self.br_wt.join()                         # layer 0's fused feed-forward sublayer reads the packed images
        Bs = self.side_windows(B, T)
        if Bs:
            # Sampler, large batch: a 64-row panel of the fused frozen-weight launch costs what its weight stream costs,
            # so 800 panels (B = 256, T = 200) are FOUR rounds of 256 workgroups for 3.125 rounds of work.  The windows
            # beyond the last full round go through the whole layer stack on the per-op row-panel kernels (which spread a
            # weight's columns over the chip) on a side branch -- forked once, joined once -- beside the full rounds.
            Bm = B - Bs
            hL = g("dt.hL", (B, T, self.d), dt)
            side = self.side_layers()
            hs = h0[Bm:]

            def side_stack():
                hh = hs
                for i, sl in enumerate(side):
                    hh = sl.forward(hh, P, out=hL[Bm:] if i + 1 == len(side) else None)
            self.br_side.run(side_stack)
            h = h0[:Bm]
            ready = aready = False
            for i, lp in enumerate(self.layers):
                h = lp.forward(h, P, out=hL[:Bm] if i + 1 == len(self.layers) else None, qkv_ready=ready, attn_ready=aready)
                ready, aready = lp.tail_done, lp.attn_tail_done
            self.br_side.join()
            h = hL
        else:
            ready = aready = False
            for lp in self.layers:
                h = lp.forward(h, P, qkv_ready=ready, attn_ready=aready)
                ready, aready = lp.tail_done, lp.attn_tail_done  # it wrote the next layer's in-projection / attention output
        out = out if out is not None else g("dt.out", (B, T, D), dt)
        if padded:
            hip.linear_fwd(h.view(M, self.d), w_out_pad, b_out_pad, out.as_strided((M, Kp), (Kp, 1)))
        elif tp:
            self.br_wt.join()                     # the padded copies of out_proj (made beside the transposes)
            hip.linear_fwd(h.view(M, self.d), tp["w_out"], tp["b_out"], out.as_strided((M, tp["Kp"]), (tp["Kp"], 1)))
        else:
            hip.linear_fwd(h.view(M, self.d), P.w("out_proj.weight"), P.v("out_proj.bias"),
                           out if out.dim() == 2 else out.view(M, D))
        self.br_wt.join()
        self.ctx = (x2, pos, h, B, T)
        return out

    def ready_order(self) -> List[str]:
        o = ["out_proj.weight", "out_proj.bias"]
        for lp in reversed(self.layers):
            o += lp.ready_order()
        return o + ["in_proj.bias"] + TimeMLPPlan.ready_order() + ["temporal_embedding.embedding.weight", "in_proj.weight"]

    def bucket_cuts(self) -> List[str]:
        """parameter names after which a gradient bucket should end: a layer's gradients become ready together (one grouped
        weight-gradient launch + one reduction), so a bucket that ends inside the NEXT layer waits a whole layer longer for
        its all-reduce than its bytes ask for"""
        return [lp.ready_order()[-1] for lp in self.layers]

    def backward(self, dout3: torch.Tensor, P: ParamSource, accumulate=False):
        x, pos, hlast, B, T = self.ctx
        M, D = x.shape
        g, dt = self.buf.get, self.dtype
        dout = dout3 if dout3.dim() == 2 else dout3.view(M, D)
        # one GPU: every split-M slab set and every per-block partial sum of a bias / LayerNorm gradient is left for the
        # optimizer launch to add up (ib_optim_step_sources, at most 64 sources): per step this removes one reduction
        # launch per weight, one per bias and one per LayerNorm
        fuse = self.fuse_reduce_into_optimizer and not accumulate and 12 * len(self.layers) + 4 <= 60
        defer, later = ([], []) if fuse else (None, None)
        # one GPU: each layer's sources are its own and its parameters are updated on its side stream right behind its
        # weight-gradient launch, beside the backward of the layers below (the optimizer is HBM traffic, those are matrix
        # work); the step's last launch then covers only the projections and the time-MLP
        early = self.early_optimizer if fuse else None
        # (the layer whose backward runs LAST keeps the step's common source lists: its range of the flat buffers borders the
        # projections' tail, so the step's last launch takes it along instead of following a launch of its own)
        last_run = self.layers[0]
        for lp in self.layers:
            lp.defer, lp.later = ([], []) if (early and lp is not last_run) else (defer, later)
        self.pending_sources = None
        def t_outproj():
            # the output projection's own gradients: three launches nothing downstream waits for -- on a side stream on one
            # GPU, so the main stream goes from the loss straight to the dgrad.  With the per-layer optimizer its parameters
            # (the head of the flat buffers) are updated right here too: the step's last launch then starts behind the layers
            op_defer = [] if early else defer
            _wgrad(self.buf, dout, hlast.view(M, self.d), P.g("out_proj.weight"), accumulate, ws_tag="dt.wso", defer=op_defer)
            P.ready("out_proj.weight")
            _colsum(self.buf, "dt.bo", dout, P.g("out_proj.bias"), accumulate)
            P.ready("out_proj.bias")
            box_op.append(op_defer)
        box_op: list = []
        ddp_overlap = bool(self.layers and self.layers[0].parent_flushes)
        outproj_pending = False
        if (fuse or ddp_overlap) and not TU.no_outproj_branch:
            # one GPU: joined with the tail's branches, before the optimizer.  Overlapped data parallel (round 5): joined in
            # front of the first flush (the first layer boundary), where its bucket goes out with the top layer's -- inline
            # its three launches (53 us, one of them the generic kernel for the 300-wide operand) sat on the critical path
            self.br_wt.run(t_outproj)
            outproj_pending = ddp_overlap
        else:
            t_outproj()                               # the flush below hands this bucket to the all-reduce
        dh = g("dt.dh", (B, T, self.d), dt)
        tp = self._tp
        if not (tp and dout.stride(0) == tp["Kp"] and dout.stride(1) == 1 and
                hip.linear_dgrad_wt(dout.as_strided((M, tp["Kp"]), (tp["Kp"], 1)), tp["w_outT"], dh.view(M, self.d))):
            hip.linear_dgrad(dout, P.w("out_proj.weight"), dh.view(M, self.d))
        if early is not None and box_op:
            # forked AGAIN from here: behind the dgrad above (which may read the weight itself) and behind the gradient
            # launches already on that branch
            self.br_wt.run(lambda: early("out_proj.", (box_op[0], None, 0, [])))
        Pin = P

        def flush_joined():
            """every flush of this backward: a flush may end a captured graph segment (a collective goes out between two
            segments), and a segment must end with every side stream joined -- the output projection's branch included"""
            nonlocal outproj_pending
            if outproj_pending:
                self.br_wt.join()
                outproj_pending = False
            Pin.flush()
        P = ParamSource(Pin.w, Pin.v, Pin.g, ready=Pin.ready, flush=flush_joined)
        if not outproj_pending:
            P.flush()                                 # (pending: its bucket goes out at the first layer boundary instead)
        prev = None                                   # (layer plan, closure, names) whose launches lag one layer
        head = None
        for lp in reversed(self.layers):
            if prev is not None:
                prev[0].branch.run(prev[1])           # beside this layer's backward
            dh = lp.backward(dh, P, accumulate, qkv_head=head)
            head = lp.pending_head                    # its in-projection dgrad is left to the next (lower) layer's launch
            if early is not None and lp is not last_run:
                # no launch issued from here on reads this layer's weights: the dgrad the layer below computes for it goes
                # through the PACKED image (refreshed at the head of the next step), the transposed copies likewise
                lp.branch.run(lambda lp=lp: early(lp.p, (lp.defer, None, 0, [(0, part.shape[1], dst, None, 1.0, part, rows)
                                                                          for part, rows, dst in lp.later])))
            lg = lp.take_lagged()                     # None unless the layers report through this plan (data parallel)
            if prev is not None:
                prev[0].branch.join()
                for nm in prev[2]:
                    P.ready(nm)
                prev = None
                if lg is None or lg[0] is not None:
                    P.flush()                         # the bucket(s) completed by the lagged layer
            if lg is not None:
                if lg[0] is None:                     # issued by the layer itself (small batches; the last layer)
                    for nm in lg[1]:
                        P.ready(nm)
                    P.flush()
                else:
                    prev = (lp, lg[0], lg[1])
        last_lag, Ptop = prev, P
        if last_lag is not None:
            # the last-run layer's grouped weight-gradient launch + reduction lag too (round 5): forked HERE, beside the
            # step's tail -- a dozen small launches on the main stream and two branches -- instead of in front of it (inline
            # it put 130 us between the last backward launch and the tail, `profiles/r05_tr_timeline_ddp.txt`).  The tail's
            # gradients are reported behind the layer's (the ready order is the flat layout), once everything is joined.
            last_lag[0].branch.run(last_lag[1])
            tail_names: List[str] = []
            P = ParamSource(Ptop.w, Ptop.v, Ptop.g, ready=tail_names.append, flush=lambda: None)
        dz0 = dh.view(M, self.d)
        w_in, gw_in = P.w("in_proj.weight"), P.g("in_proj.weight")
        # tail: three chains of small launches hang off dz0 -- the time-MLP's two halves and the frame-embedding gradients
        # -- beside the one real GEMM (the input projection's weight gradient): sibling branches
        de32 = g("dt.de32", (B, self.d), torch.float32)
        de_lp = g("dt.de_lp", (B, self.d), dt) if dt == torch.bfloat16 else None
        hip.segment_colsum(dz0, de32, seg=T, mode=0, out_bf16=de_lp)          # d e[window]

        def t_out():
            _colsum(self.buf, "dt.bi", de32, P.g("in_proj.bias"), accumulate)
            P.ready("in_proj.bias")
            self.time.backward_out_layer(de32, de_lp, P, accumulate)
        self.br_time.run(t_out)
        self.br_thid.run(lambda: self.time.backward_hidden(de32, de_lp, P, accumulate))
        gpos = P.g("temporal_embedding.embedding.weight")
        if gpos.shape[0] != T:
            raise hip.HipError(f"window length {T} != frame-embedding table rows {gpos.shape[0]}")

        def t_pos():
            dpp32 = g("dt.dpp32", (T, self.d), torch.float32)
            hip.segment_colsum(dz0, dpp32, seg=T, mode=1)                     # d posproj[frame], fp32
            # d in_proj.weight[:, D:] = dposproj^T . pos and d embedding rows = dposproj . W_p: [T, 30]-sized products
            hip.tiny_matmul(dpp32.t(), pos, gw_in[:, D:], accumulate=accumulate)
            hip.tiny_matmul(dpp32, w_in[:, D:], gpos, accumulate=accumulate)
            P.ready("temporal_embedding.embedding.weight")
        # (round 5: behind the time-MLP hidden layer's chain on ITS branch, not on a branch of its own: with the main stream,
        # layer 0's weight-gradient branch and the two time-MLP branches the tail already has four concurrent branches, a
        # captured graph runs four at a time, and the fifth -- three small launches -- only started when another had
        # finished: alone at the end of the step, `profiles/r05_tr_timeline.txt`)
        (self.br_pos if TU.pos_own_branch else self.br_thid).run(t_pos)
        _wgrad(self.buf, dz0, x, gw_in[:, :D], accumulate)
        self.br_time.join(); self.br_thid.join(); self.br_pos.join(); self.br_wt.join()
        P.ready("in_proj.weight")
        for lp in self.layers:
            lp.branch.join()
        if last_lag is not None:
            P = Ptop
            for nm in last_lag[2] + tail_names:
                P.ready(nm)
        P.flush()
        if fuse:
            self.pending_sources = (defer, None, 0, [(0, part.shape[1], dst, None, 1.0, part, rows)
                                                     for part, rows, dst in later])
