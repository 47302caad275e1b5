"""[BUILD-DEFINED] DDIM (eta = 0) sampling loop: N denoiser evaluations + updates.

One denoise step = {time-embedding gather, denoiser forward plan, DDIM update, counter++}.  The step index
lives in DEVICE memory (an int32 counter the update kernel reads), and the update kernel also writes the next
step's timestep vector, so ONE captured hipGraph of a single step can be replayed for every step of the loop
with no host involvement between steps (BASELINE config 5)."""
from typing import Optional

import torch

from .. import hip
from ..plans import ParamSource


class DDIMSampler:
    def __init__(self, model, num_sample_steps: int = 100, use_graph: bool = True):
        self.model, self.S = model, num_sample_steps
        self.use_graph = use_graph and not hip._dry_run
        self._graph: Optional[hip.Graph] = None
        self._sig = None
        self._bufs = {}

    def _step_launches(self, x, t_vec, ctr, tabs, P: ParamSource):
        plan = self.model._get_plan(x.device)
        eps_buf = self._bufs.get("eps")
        if eps_buf is not None:
            # pitched state / noise buffers [B, T, Dp] with zero pad columns (plans.infer_pitch): the plan sees row-padded
            # 2-D views; the DDIM update runs over the whole pitched buffers (0 stays 0 in the pad columns)
            B, T, Dp = x.shape
            D = self._D
            plan.forward(x.view(B * T, Dp)[:, :D], t_vec, tabs.temb, P, out=eps_buf.view(B * T, Dp)[:, :D], BT=(B, T))
            hip.ddim_step(x, eps_buf, tabs.ddim_coef, tabs.ddim_t, step_dev=ctr, t_out=t_vec)
        else:
            eps = plan.forward(x, t_vec, tabs.temb, P)
            hip.ddim_step(x, eps, tabs.ddim_coef, tabs.ddim_t, step_dev=ctr, t_out=t_vec)
        hip.counter_add(ctr, 1)

    @torch.no_grad()
    def sample(self, x_T: torch.Tensor, steps: Optional[int] = None) -> torch.Tensor:
        """x_T [B,T,D] ~ N(0,1) -> x_0.  `steps` (<= num_sample_steps) truncates the loop (benchmarks)."""
        if not x_T.is_cuda and not hip._dry_run:
            x_T = x_T.to(next(self.model.parameters()).device)
        if hip._dry_run:
            return self._sample(x_T, steps)
        # hipGraph capture is not allowed on the legacy default stream -> run on a side stream
        if getattr(self, "_stream", None) is None:
            self._stream = hip.new_stream(x_T.device)
        cur = torch.cuda.current_stream()
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            out = self._sample(x_T, steps)
        cur.wait_stream(self._stream)
        return out

    @torch.no_grad()
    def sample_noise(self, batch: int, window: int, feat: int, seed: Optional[int] = None, draw: int = 0,
                     steps: Optional[int] = None) -> torch.Tensor:
        """x_T ~ N(0,1) drawn ON the device (csrc/noise.hip: Philox keyed by (seed, draw); `seed` defaults to torch's) and
        denoised to x_0 -- no host random numbers, no H2D copy of the start state."""
        m = self.model
        dev = next(m.parameters()).device
        x_T = torch.empty((batch, window, feat), dtype=m.compute_dtype, device=dev)
        seed = int(torch.initial_seed()) if seed is None else int(seed)
        hip.diffusion_draw(seed, step=int(draw), stream_id=0x40000000, eps=x_T)
        return self.sample(x_T, steps)

    def _sample(self, x_T: torch.Tensor, steps: Optional[int] = None) -> torch.Tensor:
        m = self.model
        m.ensure_packed()
        m.sync_shadow()
        dev = m._flat.device
        tabs = m.tables(dev)
        if tabs.num_sample_steps != self.S:
            tabs.set_sampler(self.S)
        # the captured step bakes in raw pointers: the model's flat parameter buffer and bf16 shadow (a HipTrainer built
        # after a first sample() re-packs them), the schedule tables (set_sampler() of another sampler re-creates the DDIM
        # tables) -- all of them are part of the signature, so a change re-captures instead of replaying stale pointers
        shadow = m._shadow.data_ptr() if m._shadow is not None else 0
        sig = (tuple(x_T.shape), m.compute_dtype, m._flat.data_ptr(), shadow, tabs.temb.data_ptr(),
               tabs.ddim_coef.data_ptr(), tabs.ddim_t.data_ptr(), self.S)
        plan = m._get_plan(dev)
        B, T, D = x_T.shape
        Dp = plan.infer_pitch(D) if (hasattr(plan, "infer_pitch") and m.compute_dtype == torch.bfloat16) else D
        sig = sig + (Dp,)
        if sig != self._sig:
            self._sig, self._graph = sig, None
            self._bufs = {"x": torch.zeros((B, T, Dp), dtype=m.compute_dtype, device=dev),
                          "t": torch.empty(x_T.shape[0], dtype=torch.int64, device=dev),
                          "ctr": torch.zeros(1, dtype=torch.int32, device=dev)}
            if Dp != D:
                self._bufs["eps"] = torch.zeros((B, T, Dp), dtype=m.compute_dtype, device=dev)
        self._D = D
        x, t_vec, ctr = self._bufs["x"], self._bufs["t"], self._bufs["ctr"]
        x[:, :, :D].copy_(x_T.to(device=dev, dtype=m.compute_dtype))
        ctr.zero_()
        hip.fill_i64(t_vec, int(tabs.ddim_t[0]))
        P = m.param_source()
        if hasattr(plan, "set_inference"):
            plan.set_inference(True)             # weights are frozen for the whole loop
            plan.prepare_inference(P, T, D, table=tabs.temb)
        try:
            return self._loop(x, t_vec, ctr, tabs, P, steps)
        finally:
            if hasattr(plan, "set_inference"):
                plan.set_inference(False)

    def _loop(self, x, t_vec, ctr, tabs, P, steps):
        n = self.S if steps is None else min(steps, self.S)
        done = 0
        if self.use_graph and self._graph is None:
            self._step_launches(x, t_vec, ctr, tabs, P)          # eager warm-up step allocates plan buffers
            done = 1
            if n > 1:
                g = hip.Graph()
                g.begin()
                self._step_launches(x, t_vec, ctr, tabs, P)
                g.end()
                self._graph = g
        for _ in range(done, n):
            if self._graph is not None:
                self._graph.launch()
            else:
                self._step_launches(x, t_vec, ctr, tabs, P)
        return x[:, :, :self._D].clone()
