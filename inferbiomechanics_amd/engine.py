"""The fused training step and its data-parallel wiring (the MI355X-native replacement of the loop body
``zero_grad -> model(inputs) -> loss -> backward -> optimizer.step`` at src/cli/train.py:240-284 and of the
``DistributedDataParallel`` wrapper at train.py:175).

One process per GPU.  A step is a fixed sequence of C-ABI launches over buffers resident in HBM:

    [q_sample] -> plan.forward -> loss kernel (writes d loss/d out) -> plan.backward (grads straight into
    ONE flat fp32 buffer) -> bucketed RCCL all-reduce of that buffer -> ONE fused optimizer launch
    (also refreshes the bf16 weight shadow)

* no autograd, no per-parameter tensors, no host synchronisation inside a step (the loss stays on the
  device; read it with ``loss_value()`` when a report is due);
* the flat buffers are laid out in the order the backward FINISHES gradients (``plan.ready_order()``), so a
  gradient bucket is a contiguous slice that can be handed to RCCL while the rest of the backward is still
  running (comm stream + events) -- DDP's overlap without DDP's hooks;
* at steady state the launch sequence is replayed from hipGraphs (one graph per segment between two
  collectives; a single graph when world_size == 1).

Reference semantics kept: gradients are averaged over ranks (DDP mean), parameters are broadcast from rank 0
at construction (DDP ctor), optimizer arithmetic = torch.optim defaults with only lr set (train.py:183-194).
"""
from __future__ import annotations

import os
import weakref
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import hip
from ._tuning import tuning as TU
from .data.AddBiomechanicsDataset import INPUT_KEY_ORDER, LOSS_KEY_ORDER, LOSS_KEY_WIDTHS
from .loss.RegressionLossEvaluator import component_weights
from .module import HipModule, flat_layout
from .plans import ParamSource


TORCH_OPTIM_CLASS = {"sgd": "SGD", "adam": "Adam", "rmsprop": "RMSprop", "adagrad": "Adagrad", "adadelta": "Adadelta",
                     "adamax": "Adamax"}


def torch_param_groups(opt_type: str, lr: float, nparams: int) -> List[Dict]:
    """The `param_groups` list torch.optim's `optim.X(params, lr=lr)` (train.py:183-194: only lr is set) writes into its
    state dict: EVERY hyper-parameter of the class (alpha / eps / betas / momentum / weight_decay / foreach ...), taken from
    the class of the running torch, not only lr.  `Optimizer.load_state_dict` REPLACES the live groups by the saved ones,
    so a group without them makes the first `optimizer.step()` raise KeyError('alpha')."""
    cls = getattr(torch.optim, TORCH_OPTIM_CLASS[opt_type])
    g = dict(cls([torch.zeros(1)], lr=lr).state_dict()["param_groups"][0])
    g["lr"] = lr
    g["params"] = list(range(nparams))
    return [g]


def torch_state_from_flat(opt_type: str, steps: int, names, shape_of, layout, s1, s2) -> Dict:
    """per-parameter torch.optim state (`state[i] = {"step", <buffers>}`) from the flat (s1, s2) buffers.  Before the
    first step torch.optim holds NO per-parameter state (it is created lazily inside step()) -- except Adagrad, whose
    constructor creates {"step": 0, "sum": 0} and whose step() expects it."""
    keys = HipTrainer.TORCH_STATE_KEYS[opt_type]
    state = {}
    if keys and (steps > 0 or opt_type == "adagrad"):
        for i, k in enumerate(names):
            off, n = layout[k]
            st = {"step": torch.tensor(float(steps))}
            for key, buf in zip(keys, (s1, s2)):
                st[key] = buf[off:off + n].view(shape_of(k)).detach().cpu().clone()
            state[i] = st
    return state


class GradBuckets:
    """Contiguous slices of the flat gradient buffer, all-reduced (SUM) as soon as every gradient inside a
    slice has been produced.  The division by world_size is folded into the optimizer kernel."""

    def __init__(self, flat_grad: torch.Tensor, layout: "OrderedDict[str, Tuple[int, int]]", bucket_bytes: int,
                 group=None, active: Optional[bool] = None, cuts: Sequence[str] = ()):
        """cuts: parameter names after which a bucket preferably ends (a plan's layer boundaries): a bucket is closed
        there once it holds at least half of `bucket_bytes`, and otherwise as soon as it reaches `bucket_bytes`"""
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # `active`: collectives are issued.  Defaults to world > 1; the 1-rank self-test (IB_DDP_SELFTEST=1 with
        # an initialised process group) forces it so the whole comm path runs on a single GPU.
        self.active = (self.world > 1) if active is None else bool(active)
        self.on_gpu = flat_grad.is_cuda
        names = list(layout.keys())
        self.bucket_of: Dict[str, int] = {}
        self.ranges: List[Tuple[int, int]] = []
        self.members: List[List[str]] = []
        lo, cur = 0, []
        total = flat_grad.numel()
        for i, n in enumerate(names):
            cur.append(n)
            end = layout[names[i + 1]][0] if i + 1 < len(names) else total
            if (end - lo) * 4 >= bucket_bytes or i + 1 == len(names) or (n in cuts and (end - lo) * 8 >= bucket_bytes):
                for m in cur:
                    self.bucket_of[m] = len(self.ranges)
                self.ranges.append((lo, end))
                self.members.append(cur)
                lo, cur = end, []
        self._pending = [len(m) for m in self.members]
        self._works: list = []
        self._work_of: dict = {}
        # raw handles of the streams a collective has been issued from: c10d leaves completion events on them that its
        # watchdog thread polls -- such a stream must never be put into capture mode (_Recorder.begin refuses it)
        self.collective_streams: set = set()

    def reset(self):
        self._pending = [len(m) for m in self.members]
        self._works = []
        self._work_of = {}

    def mark_ready(self, name: str) -> Optional[int]:
        """returns the bucket index if `name` completed a bucket"""
        b = self.bucket_of[name]
        self._pending[b] -= 1
        if self._pending[b] < 0:
            raise hip.HipError(f"gradient of {name} reported ready twice in one step")
        return b if self._pending[b] == 0 else None

    def launch(self, b: int, inline: bool = False):
        """all-reduce bucket b as an ASYNCHRONOUS c10d collective: c10d's own communication stream first waits for
        everything enqueued on the current stream so far, the backward keeps running beside it, and finish() makes the
        current stream wait for it.  `inline` (the whole gradient in one call after the backward: nothing to overlap
        with) waits right away.  Two things measured / learnt here: an extra stream of our own around the call added two
        more event hops per bucket (51 us of idle GPU between the backward and the optimizer of a 0.25-ms step); and a
        SYNCHRONOUS collective, which c10d runs on the current stream, leaves its completion event on the very stream
        the trainer launches on -- were the graphs also CAPTURED on that stream, c10d's watchdog thread would query the
        event in the middle of a later capture and HIP refuses (the process aborts).  HipTrainer therefore captures on a
        separate stream that never carries real work; the one-bucket policy can then use the synchronous call (no hops)."""
        if not self.active:
            return
        lo, hi = self.ranges[b]
        t = self.flat[lo:hi]
        if self.on_gpu and not torch.cuda.is_current_stream_capturing():
            self.collective_streams.add(torch.cuda.current_stream().cuda_stream)
        if inline and self.on_gpu and not TU.async_inline:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)      # runs on the caller's stream: no event hops
            return
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if inline:
            w.wait()
        else:
            self._works.append(w)
            self._work_of[b] = w

    def take_work(self, b: int):
        """the outstanding Work of bucket b, handed over to the caller (who waits for it on a stream of its own: the
        per-bucket optimizer); finish() no longer waits for it"""
        w = self._work_of.pop(b, None)
        if w is not None:
            self._works = [x for x in self._works if x is not w]
        return w

    def finish(self):
        """make the compute stream wait for every outstanding bucket (a stream-level wait on the GPU path: no host block)"""
        for w in self._works:
            w.wait()
        self._works = []
        self._work_of = {}


def broadcast_parameters(flat: torch.Tensor, group=None, src: int = 0):
    """DDP-constructor semantics (train.py:175): every rank starts from rank 0's parameters."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


_drain_report = {"mode": None, "waited_s": 0.0, "polls": 0}      # the last drain of this process (tests, bench line)


def _active_c10d_works():
    """number of eager c10d Works the watchdog thread still holds, from c10d's flight recorder: every collective handed to
    the watchdog has an entry whose `retired` flag is set in the watchdog pass that reaps it (measured: 100 ms after the
    Work completed on the GPU; `state == completed` -- what the recorder's own `onlyActive` filter looks at -- comes from an
    event query and says nothing about the watchdog).  Collectives issued under capture are neither handed over nor
    recorded.  None when the recorder is off or this torch has no such call."""
    import pickle
    try:
        entries = pickle.loads(torch._C._distributed_c10d._dump_nccl_trace(True, False, False)).get("entries")
        if not entries:
            return None                              # recorder off (TORCH_FR_BUFFER_SIZE=0): nothing to read
        if "retired" not in entries[-1]:
            return None
        return sum(1 for e in entries if not e["retired"])
    except Exception:
        return None


def _drain_c10d_watchdog(device, timeout_s: float = 20.0):
    """Block until c10d's watchdog thread has reaped every eager collective issued so far.  The watchdog wakes every ~100 ms
    and polls the completion event of each Work it still holds (hipEventQuery from its own thread); the events of the eager
    warm-up steps' collectives sit on c10d's communication stream, and a poll that lands while that stream is part of a
    capture fails ("event last recorded in a capturing stream") and takes the process down.  Completed on the GPU is not
    enough -- the Work leaves the watchdog's list only in its next pass.  So: read the flight recorder until no Work is
    un-retired (deterministic); without a recorder, sleep for several watchdog periods (rounds 4 / 5a: 0.5 s was outlasted
    once on a busy host, and the `CUDAGraph.capture_begin()` wait this function used before does not exist in this torch:
    with only its 50-ms sleep one capture in ten met a watchdog pass)."""
    import time
    t0 = time.perf_counter()
    n = _active_c10d_works()
    if n is None:
        time.sleep(0.6)
        _drain_report.update(mode="sleep", waited_s=round(time.perf_counter() - t0, 3), polls=0)
        return
    polls = 1
    while n:
        if time.perf_counter() - t0 > timeout_s:
            raise hip.HipError(f"c10d's watchdog still holds {n} collective(s) after {timeout_s:.0f} s: not capturing "
                               "collectives beside them (IB_GRAPH_COLLECTIVES=0 selects the cut-graph form)")
        time.sleep(0.01)
        n = _active_c10d_works()
        polls += 1
        if n is None:                                # the recorder went away under us
            time.sleep(0.6)
            break
    _drain_report.update(mode="flight-recorder", waited_s=round(time.perf_counter() - t0, 3), polls=polls)


class _Recorder:
    """Records a step as segments: hipGraphs (captured launch runs) interleaved with host actions
    (collective launches), then replays them."""

    def __init__(self, forbidden_streams=()):
        self.actions: List[Tuple[str, object]] = []
        self._g: Optional[hip.Graph] = None
        self._forbidden = forbidden_streams

    def begin(self):
        # c10d's watchdog polls the completion events of earlier collectives; a poll that lands while the event's stream
        # is being captured aborts the process (round 1, commits 7ea3d2f / 54e1aa4).  Capture only on a stream that has
        # never carried a collective.
        if not hip._dry_run and torch.cuda.is_available() and \
                torch.cuda.current_stream().cuda_stream in self._forbidden:
            raise hip.HipError("graph capture refused: this stream has carried collectives (their completion events are "
                               "polled by c10d's watchdog thread); capture on a dedicated stream")
        self._g = hip.Graph()
        self._g.begin()

    def cut(self, host_action: Callable[[], None]):
        """close the current graph segment and schedule a host action (a collective launch) after it.
        Nothing executes while recording; replay() runs segments and actions in order."""
        self._g.end()
        self.actions.append(("graph", self._g))
        self.actions.append(("host", host_action))
        self._g = hip.Graph()
        self._g.begin()

    def end(self):
        self._g.end()
        self.actions.append(("graph", self._g))
        self._g = None

    def replay(self):
        for kind, a in self.actions:
            if kind == "graph":
                a.launch()
            else:
                a()


class HipTrainer:
    """Fused training step for one model on one GPU (+ data-parallel peers).

    task = "diffusion": batch = (x0 [B,T,D], t [B] int64, eps [B,T,D]) -> q_sample -> denoiser -> eps-MSE
    task = "regression": batch = (inputs dict of 10 keys, labels dict of 4 keys) -> FeedForwardBaseline ->
                         RegressionLossEvaluator arithmetic (component selection from `args`)
    """

    def __init__(self, model: HipModule, task: str, opt_type: str = "rmsprop", lr: float = 1e-4, args=None,
                 group=None, bucket_mb: float = 13.0, use_graph: bool = True, overlap_comm: Optional[bool] = None):
        if task not in ("diffusion", "regression"):
            raise ValueError(task)
        if opt_type not in hip.OPT:
            raise ValueError("Invalid optimizer type: " + opt_type)          # train.py:195-197
        self.model, self.task, self.opt_type, self.lr = model, task, opt_type, lr
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        import os
        # data-parallel machinery on: several ranks, or the 1-rank self-test of the comm path
        self.ddp = self.world > 1 or (os.environ.get("IB_DDP_SELFTEST") == "1" and dist.is_available()
                                      and dist.is_initialized())
        self.use_graph = use_graph and not hip._dry_run
        dev = next(model.parameters()).device
        self.device = dev
        # data parallel with hipGraphs: are the all-reduces captured INSIDE the step's graph (one graph per step) or issued
        # as host actions between graph segments?  Decided once per process by a probe in fresh child processes
        # (ddp_probe.py; IB_GRAPH_COLLECTIVES=0/1 forces the answer) -- a collective call: every rank constructs its trainer
        self.graph_collectives = False
        if self.ddp and self.use_graph and dev.type == "cuda":
            from . import ddp_probe
            self.graph_collectives = ddp_probe.decide(self.world, dist.get_rank(group), str(dist.get_backend(group)))
        self.plan = self._plan_for(dev)
        # Data-parallel policy.  Small models (gradients < 16 MiB, e.g. the MLP denoiser's 4.7 MB): the step is
        # latency-bound, the forked branches are worth more than comm/backward overlap -> branches stay ON and the
        # whole flat gradient is all-reduced in ONE call after the backward.  Large models (transformer: 52 MB):
        # buckets are all-reduced while the backward is still running (events on the main stream), branches OFF.
        nparam_bytes = sum(p.numel() for p in model.parameters()) * 4
        self.overlap_comm = self.ddp and (nparam_bytes >= (16 << 20) if overlap_comm is None else bool(overlap_comm))
        self._flush_mode = False
        if self.overlap_comm and hasattr(self.plan, "flush_each_layer"):
            # side streams stay on; completed buckets are launched at the plan's flush points (all streams joined there)
            self.plan.flush_each_layer(True)
            self._flush_mode = True
        elif self.overlap_comm and hasattr(self.plan, "branches"):
            for br in self.plan.branches():
                br.on = False               # a bucket's all-reduce is ordered after ONE stream only
        # ---- flat buffers in gradient-ready order
        order = self.plan.ready_order()
        params = OrderedDict(model.named_parameters())
        if sorted(order) != sorted(params.keys()):
            raise hip.HipError("plan.ready_order() does not cover the model's parameters")
        shapes = OrderedDict((k, tuple(params[k].shape)) for k in order)
        self.layout, total = flat_layout(shapes)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for k in order:
                off, n = self.layout[k]
                v = flat[off:off + n].view(shapes[k])
                v.copy_(params[k].data)
                params[k].data = v
        model._flat, model._layout, model._shadow = flat, self.layout, None
        self.flat = flat
        broadcast_parameters(self.flat, group)
        self.grad = torch.zeros_like(flat)
        for k in order:
            off, n = self.layout[k]
            params[k].grad = self.grad[off:off + n].view(shapes[k])
        ns = hip.OPT_NUM_STATES[opt_type]
        self.s1 = torch.zeros_like(flat) if ns >= 1 else None
        self.s2 = torch.zeros_like(flat) if ns >= 2 else None
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)      # completed steps (device-resident)
        self.ticket = torch.zeros(hip.optim_ticket_words(), dtype=torch.int32, device=dev)   # optimizer exit tickets
        self.steps_done = 0
        model.sync_shadow()
        model._shadow_fresh = True          # from here on the optimizer kernel keeps the shadow current
        self._gviews = {k: params[k].grad for k in order}
        self._params = params
        self.buckets = GradBuckets(self.grad, self.layout,
                                   int(bucket_mb * (1 << 20)) if self.overlap_comm else (1 << 62), group, active=self.ddp,
                                   cuts=set(self.plan.bucket_cuts()) if hasattr(self.plan, "bucket_cuts") else ())
        self.result = torch.zeros(64, dtype=torch.float32, device=dev)
        self.comp_w = None
        if task == "regression":
            if args is None:
                raise ValueError("regression task needs args with predict_*_components")
            self.comp_w = torch.tensor(component_weights(args), dtype=torch.float32, device=dev)
        # hipGraph capture is not allowed on the legacy default stream: the step runs on its own stream,
        # ordered after / before the caller's current stream by events
        self.stream = hip.new_stream(dev) if dev.type == "cuda" else None
        from .plans import Branch
        self._br_loss = Branch(dev, enabled=not self.overlap_comm, name="loss")
        # data parallel, bucketed: the optimizer runs PER BUCKET on a side stream as soon as that bucket's all-reduce has
        # completed (1 / world folded in), beside the backward of the layers below -- the structure of the one-GPU step's
        # per-layer optimizer; the step's last, self-counting launch then only names the buckets as done (round 5: one
        # launch over the whole flat buffer behind the last all-reduce before, 70 us on the transformer's critical path)
        # Only where completed buckets are launched at the plan's FLUSH points (layer boundaries: every launch that reads a
        # layer's weights has been issued by then -- the later dgrads go through the packed / transposed copies made at the
        # head of the step).  A plan that launches a bucket the moment its last gradient is reported may still have the
        # dgrad through those very weights ahead of it: updating them there is a race (caught by
        # tests/test_ddp_numerics_gpu.py on the MLP denoiser with forced overlap).
        self.bucket_opt = bool(self.ddp and self.overlap_comm and self._flush_mode and not TU.no_bucket_opt)
        self._br_opt = Branch(dev, enabled=self.bucket_opt, name="bucket_opt")
        self._early_done: List[Tuple[int, int]] = []
        # device-side batch draw (step_drawn): Philox key = torch's seed (torch.manual_seed(s) selects the noise as it would
        # for torch.randn), stream = data-parallel rank (ranks must not draw the same noise), step = the device counter
        self.noise_seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self.noise_stream = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
        self._static: Dict[str, torch.Tensor] = {}
        self._rec: Optional[_Recorder] = None
        self._sig = None
        self._warm = 0
        self._pinned: Dict = {}     # pointer triple -> (graph whose own slots hold it, those slots)
        self._seen: Dict = {}
        self._ready_seen: List[str] = []

    # ------------------------------------------------------------------------------------------
    def _plan_for(self, dev):
        m = self.model
        if self.task == "diffusion":
            return m._get_plan(dev)
        return m._get_plan(dev)

    def _psrc(self, cut: Optional[Callable[[int], None]] = None) -> ParamSource:
        m = self.model
        params = self._params
        if m.compute_dtype == torch.bfloat16:
            w = lambda k: m.flat_view(m._shadow, k, params[k].shape)
        else:
            w = lambda k: params[k].data
        v = lambda k: params[k].data
        g = lambda k: self._gviews[k]

        pending: List[int] = []

        def launch(b: int):
            if cut is not None:
                cut(b)
            else:
                self.buckets.launch(b)
                if self.bucket_opt:
                    self._bucket_optimizer(b)

        def ready(name: str):
            self._ready_seen.append(name)
            b = self.buckets.mark_ready(name)
            if b is not None and self.overlap_comm:
                if self._flush_mode:
                    pending.append(b)       # possibly inside a forked side stream: launched at the next flush point
                else:
                    launch(b)

        def flush():
            for b in pending:
                launch(b)
            pending.clear()
        return ParamSource(w, v, g, ready, flush)

    _slots = None          # device array {x0, eps, t} the chain / prep kernels read their batch through
    _zero_copy = False

    def _sbuf(self, name: str, like: torch.Tensor, dtype=None) -> torch.Tensor:
        dtype = dtype or like.dtype
        key = (name, tuple(like.shape), dtype)
        t = self._static.get(key)
        if t is None:
            t = torch.empty(like.shape, dtype=dtype, device=self.device)
            self._static[key] = t
        return t

    def _in_size(self, frames: int) -> int:
        """width of the regression model's input matrix: fixed by the feedforward ctor, per window length for Groundlink"""
        m = self.model
        return m.input_size_for(frames) if hasattr(m, "input_size_for") else m.input_size

    def _out_frames(self, frames: int) -> int:
        m = self.model
        return m.output_frames_for(frames) if hasattr(m, "output_frames_for") else m.num_output_frames

    # ---- the launch sequence ---------------------------------------------------------------------
    def _launches(self, st: Dict[str, torch.Tensor], cut=None):
        m, plan, dt = self.model, self.plan, self.model.compute_dtype
        self.buckets.reset()
        self._ready_seen = []
        self._early_done = []
        P = self._psrc(cut)
        if self.task == "diffusion":
            x0, t, eps = st["x0"], st["t"], st["eps"]
            tabs = m.tables(self.device)
            B, T, D = x0.shape
            M = B * T
            if "draw" in st:
                # the batch is made on the device by ONE launch: x0 gathered out of the HBM window table (or already staged),
                # t and eps drawn from the counter-based generator keyed by the device-resident step counter -- a replayed
                # graph draws fresh numbers every step
                mc = self._mcache if "widx" in st else None
                hip.diffusion_draw(self.noise_seed, step_dev=self.step_dev, stream_id=self.noise_stream, eps=eps, t=t,
                                   num_train_steps=m.num_train_steps, table=None if mc is None else mc.table,
                                   idx=st.get("widx"), x0=x0 if mc is not None else None)
            if hasattr(plan, "chain_ok") and plan.chain_ok(D):
                plan.fuse_reduce_into_optimizer = not self.ddp and not TU.no_opt_fuse
                # MLP denoiser, bf16: q_sample + forward + loss + the dgrad chain are ONE launch (csrc/chain.hip)
                if self._slots is None:
                    self._slots = torch.zeros(4, dtype=torch.int64, device=self.device)
                    hip.set_ptrs(self._slots, [x0, eps, t])
                plan.chain_step(x0, eps, t, tabs, P, self.result, slots=self._slots)
                self._zero_copy = plan.slots_used        # later steps consume device-resident batches in place
                return self._finish_step(cut)
            # activations that are D (= 300) wide live in buffers with a 16-byte-aligned row pitch (304): the GEMM
            # operand pieces are then aligned 16-byte loads (the unpadded rows were 8-byte aligned: +11 us per wgrad)
            if hasattr(plan, "fuse_reduce_into_optimizer"):
                plan.fuse_reduce_into_optimizer = not self.ddp and not TU.no_opt_fuse
            if hasattr(plan, "early_optimizer"):
                # one GPU: a layer's parameters are updated as soon as its gradient is complete, on the layer's side stream
                # beside the backward of the layers below, instead of in one launch at the end of the step
                early = plan.fuse_reduce_into_optimizer and not TU.no_early_opt
                if early:
                    # probed HERE, before the first launch of the step: a layer whose parameters are not one aligned range
                    # of the flat buffer falls back to the single end-of-step launch instead of failing mid-step
                    try:
                        for lp in getattr(plan, "layers", ()):
                            self._prefix_range(lp.p)
                        self._prefix_range("out_proj.")
                    except hip.HipError:
                        early = False
                plan.early_optimizer = self._early_optimizer if early else None
                self._early_done = []
            if hasattr(plan, "train_pitch"):
                Dp = plan.train_pitch(D, M)        # 320 for the transformer denoiser at training batch sizes (plans._train_pad)
            else:
                Dp = D if TU.no_pad else (D + 7) // 8 * 8
            # zeroed once at creation: the pad columns are operands of the padded projections and never written otherwise
            xt = plan.buf.get("tr.xt", (M, Dp), dt, zero=True)[:, :D]
            pred = plan.buf.get("tr.pred", (M, Dp), dt, zero=True)[:, :D]
            dpred = plan.buf.get("tr.dpred", (M, Dp), dt, zero=True)[:, :D]
            hip.q_sample(x0, eps, t, tabs.sqrt_ab, tabs.sqrt_1mab, xt)
            plan.forward(xt, t, tabs.temb, P, out=pred, BT=(B, T))
            ws = plan.buf.bytes("tr.mse", hip.mse_loss_workspace_bytes(M * D))
            hip.mse_loss_partial(pred, eps, ws, dpred=dpred)          # dL/dpred + per-block partial sums
            n = M * D
            self._br_loss.run(lambda: hip.mse_loss_finalize(ws, self.result, n))   # the scalar: off the chain
            plan.backward(dpred, P, accumulate=False)
            self._br_loss.join()
            P.flush()                                  # buckets completed after the plan's last flush point
        else:
            B = st["lab0"].shape[0]
            frames_in = self._cache.frames if "widx" in st else st["in0"].shape[1]
            x = plan.buf.get("ff.x", (B, self._in_size(frames_in)), dt)
            if "widx" in st:
                # cached windows: ONE gather launch fills the model input and the four label tensors
                hip.gather_windows(self._cache.table, st["widx"], x, [st[f"lab{i}"] for i in range(4)])
            else:
                hip.concat_keys([st[f"in{i}"] for i in range(len(INPUT_KEY_ORDER))], x)
            if hasattr(plan, "fuse_reduce_into_optimizer"):
                plan.fuse_reduce_into_optimizer = not self.ddp and not TU.no_opt_fuse
            if hasattr(m, "output_frames_for"):
                # Groundlink: [B, F', 30] with the four outputs interleaved per frame; the dropout masks are keyed on
                # the device-resident step counter, so a replayed graph draws fresh masks every step
                out = plan.forward(x, P, training=m.training, step_dev=self.step_dev)
            elif getattr(m, "train_mode_matters", False):
                # feedforward model with --dropout / --batchnorm: same device-resident step counter for the masks; the
                # BatchNorm running statistics are updated in place by the captured launches
                out = plan.forward(x, P, training=m.training, step_dev=self.step_dev)
            else:
                out = plan.forward(x, P)
            F = self._out_frames(frames_in)
            views = m.split_output(out)
            outs = tuple(views[k] for k in LOSS_KEY_ORDER)
            G = plan.buf.get("tr.dout", out.shape, dt)
            gv = m.split_output(G)
            grads = tuple(gv[k] for k in LOSS_KEY_ORDER)
            labs = tuple(st[f"lab{i}"] for i in range(4))
            ws = plan.buf.bytes("tr.rl", hip.regression_loss_workspace_bytes(B, F))
            hip.regression_loss(outs, labs, self.comp_w, self.result, ws, grads=grads, threshold=10.0)
            plan.backward(G, P, accumulate=False)
            P.flush()
        self._finish_step(cut)

    def _finish_step(self, cut):
        m, dt = self.model, self.model.compute_dtype
        if self.ddp:
            if cut is not None:
                cut(-1 if self.overlap_comm else -2)
            elif not self.overlap_comm:
                self.buckets.launch(0, inline=True)  # one bucket = the whole flat gradient, after all joins
            else:
                self._br_opt.join()                  # the per-bucket optimizer launches (each waited for its own all-reduce)
                self.buckets.finish()
        # self-counting optimizer launch: uses *step_dev + 1 and publishes it itself (no separate counter launch)
        src = getattr(self.plan, "pending_sources", None)
        done, self._early_done = list(self._early_done), []
        if src is not None:
            self.plan.pending_sources = None
        elif done:
            src = ([], None, 0, [], [])               # data parallel, per-bucket optimizer: nothing is left to sum
        # The launch covers only what the earlier launches of this step left: ranges already updated at the head / tail of
        # the flat buffers are cut off (a launch over the whole 13-M-element buffer whose blocks find most of it done was
        # 18 us at the very end of the transformer step; over the 1 M elements that are left it is 4), ranges in the middle
        # are named as done (source kind 3).  Everything updated already: one block that only counts the step.
        lo, hi, inner = self._last_launch_range(done, src)
        if src is not None and (inner or len(src) > 4):
            src = tuple(src[:4]) + ([self.grad[a:b] for a, b in inner],)
        sl = lambda t: None if t is None else t[lo:hi]
        hip.optim_step(self.opt_type, self.flat[lo:hi], self.grad[lo:hi], sl(self.s1), sl(self.s2), self.lr, step=0,
                       step_dev=self.step_dev, ticket=self.ticket, grad_scale=1.0 / self.world,
                       shadow=sl(m._shadow) if dt == torch.bfloat16 else None, sources=src)

    def _last_launch_range(self, done, src):
        """[lo, hi) of the flat buffers for the step's last optimizer launch + the done ranges left inside it"""
        n = self.flat.numel()
        merged: List[List[int]] = []
        for a, b in sorted(done):
            if merged and a <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], b)
            else:
                merged.append([a, b])
        lo, hi = 0, n
        if merged and merged[0][0] == 0:
            lo = merged.pop(0)[1]
        if merged and merged[-1][1] == n:
            hi = merged.pop()[0]
        if lo >= hi:                                  # nothing left: a one-block launch that publishes the step number
            return n - 64, n, [(n - 64, n)]
        if src is not None:                           # every gradient source of this launch must lie inside the range
            g0 = self.grad.data_ptr()
            dsts = [dw for _, _, dw in src[0]] + [seg[2] for seg in src[3]] + [seg[3] for seg in src[3] if seg[3] is not None]
            for t in dsts:
                off = (t.data_ptr() - g0) // 4
                if 0 <= off < n and not (lo <= off and off + t.numel() <= hi):
                    return 0, n, [(a, b) for a, b in sorted(done)]
        return lo, hi, [(a, b) for a, b in merged]

    def _bucket_optimizer(self, b: int, issue: bool = True, mark: bool = True):
        """the optimizer over bucket b's range of the flat buffers, on the `bucket_opt` side stream behind that bucket's
        all-reduce: same step number as the step's last, self-counting launch (*step_dev + 1), which skips the range.
        `mark` without `issue`: recording a cut-graph step (the range is named as done in the captured last launch; the
        launch itself is a host action of the replay, `issue` without `mark`)."""
        lo, hi = self.buckets.ranges[b]
        if mark:
            self._early_done.append((lo, hi))
        if not issue:
            return
        m, dt = self.model, self.model.compute_dtype
        sl = lambda t: None if t is None else t[lo:hi]
        w = self.buckets.take_work(b)

        def fn():
            if w is not None:
                w.wait()                             # stream-level on the GPU path: the SIDE stream waits, not the backward
            hip.optim_step(self.opt_type, self.flat[lo:hi], self.grad[lo:hi], sl(self.s1), sl(self.s2), self.lr, step=1,
                           step_dev=self.step_dev, grad_scale=1.0 / self.world,
                           shadow=sl(m._shadow) if dt == torch.bfloat16 else None)
        self._br_opt.run(fn)

    def _prefix_range(self, prefix: str) -> Tuple[int, int]:
        """[lo, hi) of the flat buffers holding exactly the parameters whose names start with `prefix`"""
        cache = self.__dict__.setdefault("_prefix_ranges", {})
        if prefix not in cache:
            names = list(self.layout.keys())
            idx = [i for i, k in enumerate(names) if k.startswith(prefix)]
            if not idx or idx != list(range(idx[0], idx[-1] + 1)):
                raise hip.HipError(f"parameters of '{prefix}' are not one contiguous range of the flat buffer")
            lo = self.layout[names[idx[0]]][0]
            hi = self.layout[names[idx[-1] + 1]][0] if idx[-1] + 1 < len(names) else self.flat.numel()
            if lo % 4 or hi % 4:
                raise hip.HipError("flat ranges must be 16-byte aligned")
            cache[prefix] = (lo, hi)
        return cache[prefix]

    def _early_optimizer(self, prefix: str, sources):
        """the optimizer over ONE layer's range (issued by the plan on that layer's side stream once its gradient sources
        are complete): same step number as the step's last, self-counting launch (*step_dev + 1), which skips the range"""
        lo, hi = self._prefix_range(prefix)
        m, dt = self.model, self.model.compute_dtype
        sl = lambda t: None if t is None else t[lo:hi]
        hip.optim_step(self.opt_type, self.flat[lo:hi], self.grad[lo:hi], sl(self.s1), sl(self.s2), self.lr, step=1,
                       step_dev=self.step_dev, grad_scale=1.0 / self.world,
                       shadow=sl(m._shadow) if dt == torch.bfloat16 else None, sources=sources)
        self._early_done.append((lo, hi))

    def _stage(self, batch) -> Dict[str, torch.Tensor]:
        """copy the batch into the static input buffers the (captured) launch sequence reads"""
        dt = self.model.compute_dtype
        st: Dict[str, torch.Tensor] = {}
        if self.task == "diffusion" and isinstance(batch[0], str):
            # ("motion", cache, idx): x0 from the HBM window table; ("draw", x0): x0 from the caller -- t / eps drawn on the
            # device in both.  The launch sequence reads the trainer's own static buffers (fixed addresses: the pointer
            # slots of the chain path never change, so the step replays from a pinned graph without a pointer launch).
            if batch[0] == "motion":
                _, cache, idx = batch
                if cache.table.dtype != dt:
                    raise hip.HipError(f"motion cache holds {cache.table.dtype}, the model computes in {dt}")
                if self._mcache is not cache:
                    self._mcache, self._rec, self._sig = cache, None, None          # another table: re-capture
                B, T, D = idx.numel(), cache.window, cache.feat
                b = self._sbuf("widx", idx, torch.int64)
                b.copy_(idx, non_blocking=True)
                st["widx"] = b
                st["x0"] = self._sbuf("x0", torch.empty((B, T, D), device="meta"), dt)
            else:
                x0 = batch[1]
                st["x0"] = self._sbuf("x0", x0, dt)
                st["x0"].copy_(x0, non_blocking=True)
            st["eps"] = self._sbuf("eps", st["x0"], dt)
            st["t"] = self._sbuf("t", torch.empty((st["x0"].shape[0],), device="meta"), torch.int64)
            st["draw"] = st["t"]
            self._srcs = [st["x0"], st["eps"], st["t"]]
            self._last_drawn = (st["x0"], st["t"], st["eps"])
        elif self.task == "diffusion":
            x0, t, eps = batch
            srcs = []
            for name, src, d in (("x0", x0, dt), ("eps", eps, dt), ("t", t, torch.int64)):
                b = self._sbuf(name, src, d)
                # chain path: the kernels read the batch through device pointer slots, so a batch that already lies
                # in HBM in the right dtype is consumed in place (the staging copies were ~15 us of a 0.25 ms step)
                inplace = self._zero_copy and src.is_cuda and src.device == b.device and src.dtype == d \
                    and src.is_contiguous() and src.data_ptr() % 16 == 0 and not TU.no_zero_copy
                if not inplace:
                    b.copy_(src, non_blocking=True)
                srcs.append(src if inplace else b)
                st[name] = b
            self._srcs = srcs
        elif isinstance(batch, tuple) and len(batch) == 3 and batch[0] == "windows":
            _, cache, idx = batch
            if cache.x_elems != self._in_size(cache.frames) or cache.out_frames != self._out_frames(cache.frames):
                raise hip.HipError(f"window cache geometry (x {cache.x_elems}, F' {cache.out_frames}) does not match the "
                                   f"model (input {self._in_size(cache.frames)}, F' {self._out_frames(cache.frames)})")
            if self._cache is not cache:
                self._cache, self._rec, self._sig = cache, None, None      # another table: re-capture
            b = self._sbuf("widx", idx, torch.int64)
            b.copy_(idx, non_blocking=True)
            st["widx"] = b
            for i, shp in enumerate(cache.label_shapes(idx.numel())):
                st[f"lab{i}"] = self._sbuf(f"lab{i}", torch.empty(shp, device="meta"), torch.float32)
        else:
            inputs, labels = batch
            for i, k in enumerate(INPUT_KEY_ORDER):
                b = self._sbuf(f"in{i}", inputs[k], torch.float32)
                b.copy_(inputs[k], non_blocking=True)
                st[f"in{i}"] = b
            for i, k in enumerate(LOSS_KEY_ORDER):
                b = self._sbuf(f"lab{i}", labels[k], torch.float32)
                b.copy_(labels[k], non_blocking=True)
                st[f"lab{i}"] = b
        return st

    def step(self, batch) -> torch.Tensor:
        """one fused training step; returns the DEVICE scalar holding this step's loss (no sync)."""
        if self.stream is None:
            return self._step(batch)
        cur = torch.cuda.current_stream()
        if cur == self.stream:          # the caller already works on the trainer's stream (`with torch.cuda.stream(trainer.stream)`
            return self._step(batch)    # around its loop): no event hand-over between two hardware queues per step
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            r = self._step(batch)
        cur.wait_stream(self.stream)
        return r

    def adopt_stream(self):
        """Make the trainer's stream the CURRENT torch stream of this thread (returns the previous one, for
        `torch.cuda.set_stream(prev)` afterwards).  A loop that calls step() from another stream pays two event hand-overs
        between hardware queues per step (measured: 25 us of a 0.218 ms MLP-denoiser step, 50 us of a 0.19 ms feedforward
        step); a training loop that does all of its device work on the trainer's stream pays none.  cli/train.py and
        bench.py do this."""
        if self.stream is None:
            return None
        prev = torch.cuda.current_stream()
        self.stream.wait_stream(prev)           # whatever the caller enqueued so far is ordered before the first step
        torch.cuda.set_stream(self.stream)
        return prev

    _cache = None

    def step_windows(self, cache, idx: torch.Tensor) -> torch.Tensor:
        """one fused regression step over windows `idx` (int64, on the device) of a `data.WindowCache.DeviceWindowCache`:
        the batch is gathered on the device by one launch -- no host-side window assembly, no collate, no H2D copy"""
        if self.task != "regression":
            raise hip.HipError("step_windows: the window cache feeds the regression models")
        return self.step(("windows", cache, idx))

    _mcache = None
    _last_drawn = None

    def step_drawn(self, cache, idx: torch.Tensor) -> torch.Tensor:
        """one fused diffusion step over windows `idx` (int64, on the device) of a `data.WindowCache.DeviceMotionCache`:
        x0 is gathered and that step's timesteps / noise are drawn ON the device by one launch (`ib_diffusion_draw`) ahead
        of the step -- no host random numbers, no H2D copy.  The draw is a pure function of (torch seed, completed steps,
        rank, element), so a run is reproducible and ranks draw different noise."""
        if self.task != "diffusion":
            raise hip.HipError("step_drawn: the motion cache feeds the diffusion models")
        return self.step(("motion", cache, idx))

    def step_x0(self, x0: torch.Tensor) -> torch.Tensor:
        """as step_drawn for windows the caller supplies (a DataLoader batch [B, T, D], host or device memory)"""
        if self.task != "diffusion":
            raise hip.HipError("step_x0: diffusion models only")
        return self.step(("draw", x0))

    def drawn_batch(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(x0, t, eps) of the LAST step_drawn / step_x0 call, as the device left them (tests; dev-set evaluation)"""
        if self._last_drawn is None:
            raise hip.HipError("drawn_batch: no step_drawn / step_x0 call yet")
        return self._last_drawn

    _srcs = None            # this step's {x0, eps, t} tensors behind the pointer slots (chain path), else None
    _cap_stream = None
    MAX_PINNED_GRAPHS = 128
    MAX_DICT_CAPTURES = 16      # in-run captures of dict-batch graphs (each ~1 ms of host work in the middle of a run)
    _dict_captures = 0

    captures = 0            # graphs captured so far (bench.py asserts that none falls inside its timed region)

    def _capture(self, st) -> "_Recorder":
        self.captures += 1
        rec = _Recorder(self.buckets.collective_streams)
        # graph_collectives (ddp_probe.py's verdict, or IB_GRAPH_COLLECTIVES=1): the all-reduces are CAPTURED (c10d enqueues
        # the RCCL kernels on its communication stream, which joins the capture through the event edges it records): one graph
        # per step, no graph cut and no host work per collective.  The default wherever the start-up probe -- the same two
        # forms run side by side in fresh child processes on the job's own ranks -- finds that it works.
        # A hazard of the captured form: on the bucketed (overlap_comm) path c10d's own communication stream joins the capture, and that
        # stream carried the eager warm-up steps' collectives whose completion events the c10d watchdog polls from its own
        # thread -- a poll that lands while the stream captures aborts the process (the hazard _Recorder.begin refuses for
        # the caller's streams; it cannot see c10d's).  So before capturing: wait on every outstanding Work, drain the
        # device, and wait until the watchdog has reaped them all (_drain_c10d_watchdog); afterwards nothing it still polls
        # sits on that stream.
        graph_collectives = self.ddp and self.graph_collectives
        if graph_collectives:
            self.buckets.finish()
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
                _drain_c10d_watchdog(self.device)

        def cut(b: int):
            if graph_collectives:
                if b >= 0:
                    self.buckets.launch(b)
                    if self.bucket_opt:
                        self._bucket_optimizer(b)          # a parallel branch of the graph behind the captured all-reduce
                elif b == -1:
                    self._br_opt.join()
                    self.buckets.finish()
                else:
                    self.buckets.launch(0, inline=True)
                return
            if b >= 0:
                if self.bucket_opt:
                    # the collective and the bucket's optimizer launch are host actions of the replay (eager, between two
                    # graph segments); the captured last launch names the range as done
                    self._bucket_optimizer(b, issue=False)
                    rec.cut(lambda: (self.buckets.launch(b), self._bucket_optimizer(b, mark=False)))
                else:
                    rec.cut(lambda: self.buckets.launch(b))
            elif b == -1:
                rec.cut(lambda: (self._br_opt.join(), self.buckets.finish()))
            else:
                rec.cut(lambda: self.buckets.launch(0, inline=True))
        # Captured on a stream of its own, launched on the trainer's stream.  c10d's watchdog thread polls the completion
        # events of earlier collectives; an event that sits on a stream WHILE that stream is being captured makes the
        # poll fail and aborts the process.  The capture stream carries no real work, so no such event ever sits on it.
        if self._cap_stream is None and self.device.type == "cuda":
            self._cap_stream = hip.new_stream(self.device)
        if self._cap_stream is not None:
            with torch.cuda.stream(self._cap_stream):
                rec.begin()
                self._launches(st, cut=cut if self.ddp else None)
                rec.end()
        else:
            rec.begin()
            self._launches(st, cut=cut if self.ddp else None)
            rec.end()
        return rec

    def _dict_batch_direct(self, batch):
        """regression step over a dict batch whose 14 tensors already lie in HBM as contiguous fp32: (pointer key, the
        launch sequence's inputs = those very tensors); None when any of them would need a staging copy"""
        if self.task != "regression" or not isinstance(batch, tuple) or len(batch) != 2 or not isinstance(batch[0], dict):
            return None
        inputs, labels = batch
        st, ptrs = {}, []
        for pre, keys, src in (("in", INPUT_KEY_ORDER, inputs), ("lab", LOSS_KEY_ORDER, labels)):
            for i, k in enumerate(keys):
                t = src.get(k)
                if not isinstance(t, torch.Tensor) or t.device != self.device or t.dtype != torch.float32 \
                        or not t.is_contiguous() or t.data_ptr() % 16:
                    return None
                st[f"{pre}{i}"] = t
                ptrs.append(t.data_ptr())
        return tuple(ptrs), st

    def _step(self, batch) -> torch.Tensor:
        self._srcs = None
        # Dict batches that recur at the same addresses (a loader recycling its device buffers; a benchmark loop): from the
        # second appearance such a batch gets a graph of its own that reads the caller's tensors where they lie -- the 14
        # staging copies per step (10 input keys + 4 labels, each a tiny D2D launch issued by the host: ~45 us of the 0.12-ms
        # fp32 reference-shape step) disappear.  Anything else is staged into the static buffers as before.
        if self.use_graph and self._rec is not None and not TU.no_pinned_graphs \
                and not (self.ddp and self.graph_collectives):
            d = self._dict_batch_direct(batch)
            if d is not None:
                key, st_d = d
                sig_d = tuple((k, tuple(v.shape)) for k, v in st_d.items()) + (("training", bool(self.model.training)),)
                if sig_d == self._sig:
                    # Only the SAME tensor objects coming back qualify (held by weak references): an ordinary loader makes
                    # fresh `.to(device)` tensors every step and the caching allocator recycles their addresses -- an address
                    # tuple seen twice is then no sign of a recycled buffer, and a graph pinned on it would keep 14 dead
                    # tensors (and itself) alive for nothing.  A pinned entry whose tensors have died is dropped.
                    tensors = list(st_d.values())
                    same = lambda refs: len(refs) == len(tensors) and all(r() is t for r, t in zip(refs, tensors))
                    pin = self._pinned.get(key)
                    if pin is not None and not same(pin[2]):
                        del self._pinned[key]
                        pin = None
                    if pin is None and len(self._pinned) < self.MAX_PINNED_GRAPHS \
                            and self._dict_captures < self.MAX_DICT_CAPTURES:
                        seen = self._seen.get(key)
                        if seen is not None and same(seen):
                            self._dict_captures += 1
                            pin = self._pinned[key] = (self._capture(st_d), None, seen)
                            del self._seen[key]
                        elif len(self._seen) < 256 or key in self._seen:
                            self._seen[key] = [weakref.ref(t) for t in tensors]
                    if pin is not None:
                        pin[0].replay()
                        self.steps_done += 1
                        return self.result[0]
        st = self._stage(batch)
        # model.training is baked into a captured graph (Groundlink / dropout layers choose their launches by it)
        sig = tuple((k, tuple(v.shape)) for k, v in st.items()) + (("training", bool(self.model.training)),)
        if sig != self._sig:
            self._sig, self._rec, self._warm = sig, None, 0
            self._pinned, self._seen = {}, {}
        slots_live = self._slots is not None and self._srcs is not None
        key = tuple(t.data_ptr() for t in self._srcs) if slots_live else None
        if self._rec is not None:
            # A loader that recycles its device buffers (a prefetch ring; the staging buffers themselves) shows the same
            # pointer triple again and again: from its second appearance a triple gets a graph of its own whose slots hold
            # those pointers for good -- no per-step pointer-update launch ahead of the graph (~5 us of a 0.24 ms step).
            pin = self._pinned.get(key) if key is not None else None
            if pin is None and key is not None and len(self._pinned) < self.MAX_PINNED_GRAPHS \
                    and not TU.no_pinned_graphs:
                n = self._seen.get(key, 0) + 1
                if len(self._seen) < 4096 or key in self._seen:
                    self._seen[key] = n
                if n >= 2:
                    pin = self._pinned[key] = self._pin(st)
            if pin is not None:
                pin[0].replay()
            else:
                if slots_live:
                    hip.set_ptrs(self._slots, self._srcs)
                self._rec.replay()
        elif not self.use_graph or self._warm < 2:
            if slots_live:
                hip.set_ptrs(self._slots, self._srcs)
            self._launches(st)                  # eager (warm-up allocates every plan buffer)
            if self._warm == 0 and self._ready_seen != list(self.layout.keys()):
                raise hip.HipError("plan.backward() did not report gradients in ready_order()")
            self._warm += 1
        else:
            if slots_live:
                hip.set_ptrs(self._slots, self._srcs)
            self._rec = self._capture(st)
            self._rec.replay()                  # the capture itself executed nothing
        self.steps_done += 1
        return self.result[0]

    def pin_batches(self, batches) -> int:
        """Capture AHEAD OF TIME everything a run over `batches` will replay: the generic graph and -- chain path, batches
        already resident in HBM in the compute dtype -- the graph of every distinct batch-pointer triple (otherwise a
        triple gets its graph on its second appearance, i.e. ~1 ms of capture in the middle of the run).  A loader that
        recycles a ring of device buffers calls this once with the ring.  Returns the number of graphs captured here."""
        if self.stream is None or not self.use_graph:
            return 0
        before = self.captures
        while self._rec is None:                 # two eager warm-up steps, then the generic capture
            self.step(batches[0])
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for batch in batches:
                self._srcs = None
                st = self._stage(batch)
                if self._slots is None or self._srcs is None or TU.no_pinned_graphs:
                    break
                key = tuple(t.data_ptr() for t in self._srcs)
                if key in self._pinned or len(self._pinned) >= self.MAX_PINNED_GRAPHS:
                    continue
                self._pinned[key] = self._pin(st)
        cur.wait_stream(self.stream)
        return self.captures - before

    def _pin(self, st):
        own = torch.zeros(4, dtype=torch.int64, device=self.device)
        hip.set_ptrs(own, self._srcs)
        generic, self._slots = self._slots, own
        try:
            return (self._capture(st), own)
        finally:
            self._slots = generic

    def loss_value(self) -> float:
        """host readback of the last step's loss (synchronises)"""
        return float(self.result[0].cpu())

    # ---- checkpoint payload (grammar of train.py:272-278) ------------------------------------------
    # torch.optim state names behind the flat (s1, s2) buffers (csrc/optim.hip: the order of its update formulas)
    TORCH_STATE_KEYS = {"sgd": (), "adam": ("exp_avg", "exp_avg_sq"), "rmsprop": ("square_avg",), "adagrad": ("sum",),
                        "adadelta": ("square_avg", "acc_delta"), "adamax": ("exp_avg", "exp_inf")}

    def optimizer_state_dict(self) -> Dict:
        return {"opt_type": self.opt_type, "lr": self.lr, "step": self.steps_done,
                "layout": {k: list(v) for k, v in self.layout.items()},
                "s1": None if self.s1 is None else self.s1.detach().cpu(),
                "s2": None if self.s2 is None else self.s2.detach().cpu()}

    def torch_optimizer_state_dict(self) -> Dict:
        """the same state in torch.optim's own grammar ({'state': {i: {...}}, 'param_groups': [...]}, parameters numbered
        in model.parameters() order as `optim.X(model.parameters(), lr=...)` numbers them, train.py:183-194): what a
        reference-style (`--eager` / torch.optim) run or the reference itself can load"""
        names = [k for k, _ in self.model.named_parameters()]
        state = torch_state_from_flat(self.opt_type, self.steps_done, names, lambda k: self._params[k].shape, self.layout,
                                      self.s1, self.s2)
        return {"state": state, "param_groups": torch_param_groups(self.opt_type, self.lr, len(names))}

    def load_optimizer_state_dict(self, sd: Dict):
        """accepts this trainer's flat payload AND a torch.optim state dict (a checkpoint written by the reference, by
        `train --eager`, or by torch_optimizer_state_dict())"""
        if "state" in sd and "param_groups" in sd:
            return self._load_torch_optimizer_state(sd)
        if sd.get("opt_type") != self.opt_type or {k: list(v) for k, v in self.layout.items()} != sd.get("layout"):
            raise hip.HipError("optimizer state does not match this trainer (optimizer type or parameter layout)")
        if self.s1 is not None:
            self.s1.copy_(sd["s1"])
        if self.s2 is not None:
            self.s2.copy_(sd["s2"])
        self.steps_done = int(sd["step"])
        self.step_dev.fill_(self.steps_done)

    def _load_torch_optimizer_state(self, sd: Dict):
        names = [k for k, _ in self.model.named_parameters()]
        ids = [i for grp in sd["param_groups"] for i in grp["params"]]
        if len(ids) != len(names):
            raise hip.HipError(f"torch.optim state covers {len(ids)} parameters, the model has {len(names)}")
        keys = self.TORCH_STATE_KEYS[self.opt_type]
        steps = 0
        for buf in (self.s1, self.s2):
            if buf is not None:
                buf.zero_()
        for pid, k in zip(ids, names):
            st = sd["state"].get(pid)
            if not st:
                continue                      # never stepped (or plain SGD: stateless)
            missing = [key for key in keys if key not in st]
            if missing:
                raise hip.HipError(f"torch.optim state of '{k}' lacks {missing}: not a {self.opt_type} state")
            off, n = self.layout[k]
            for key, buf in zip(keys, (self.s1, self.s2)):
                if tuple(st[key].shape) != tuple(self._params[k].shape):
                    raise hip.HipError(f"torch.optim state '{key}' of '{k}' has shape {tuple(st[key].shape)}")
                buf[off:off + n].view(self._params[k].shape).copy_(st[key].to(torch.float32))
            steps = max(steps, int(float(st.get("step", 0))))
        self.steps_done = steps
        self.step_dev.fill_(steps)

    def refresh_after_param_load(self):
        """call after model.load_state_dict(): re-cast the bf16 shadow"""
        self.model._shadow_fresh = False
        self.model.sync_shadow()
        self.model._shadow_fresh = True
