"""torch.optim <-> flat optimizer state (engine.torch_param_groups / torch_state_from_flat, used by
HipTrainer.torch_optimizer_state_dict and cli.abstract_command.flat_to_torch_optimizer_state): a state dict built from the
fused trainer's flat buffers must not only LOAD into `torch.optim.X(params, lr=...)` (train.py:183-194) -- the optimizer must
also STEP afterwards and continue the trajectory.  `Optimizer.load_state_dict` replaces the live param_groups by the saved
ones: groups that carry only lr / params raise KeyError('alpha' / 'betas' / ...) at the first step (round-2 advisor finding)."""
import pytest
import torch

from inferbiomechanics_amd.engine import HipTrainer, TORCH_OPTIM_CLASS, torch_param_groups, torch_state_from_flat
from inferbiomechanics_amd.module import flat_layout

SHAPES = {"a.weight": (5, 7), "a.bias": (5,), "b.weight": (3, 5)}


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(s, generator=g) for k, s in SHAPES.items()}


def _grads(seed):
    g = torch.Generator().manual_seed(100 + seed)
    return {k: torch.randn(s, generator=g) for k, s in SHAPES.items()}


@pytest.mark.parametrize("opt_type", list(TORCH_OPTIM_CLASS))
@pytest.mark.parametrize("steps_before", [0, 2])
def test_state_built_from_flat_buffers_loads_and_steps(opt_type, steps_before):
    cls = getattr(torch.optim, TORCH_OPTIM_CLASS[opt_type])
    names = list(SHAPES)
    pa = {k: v.clone().requires_grad_(True) for k, v in _params(0).items()}
    A = cls(list(pa.values()), lr=1e-2)
    for i in range(steps_before):
        for k, g in _grads(i).items():
            pa[k].grad = g.clone()
        A.step()
    # what HipTrainer holds: flat s1 / s2 in its layout
    layout, total = flat_layout({k: tuple(s) for k, s in SHAPES.items()})
    keys = HipTrainer.TORCH_STATE_KEYS[opt_type]
    bufs = [torch.zeros(total) for _ in keys]
    sa = A.state_dict()["state"]
    for i, k in enumerate(names):
        off, n = layout[k]
        for key, buf in zip(keys, bufs):
            if i in sa and key in sa[i]:
                buf[off:off + n] = sa[i][key].reshape(-1)
    s1 = bufs[0] if len(bufs) > 0 else None
    s2 = bufs[1] if len(bufs) > 1 else None
    sd = {"state": torch_state_from_flat(opt_type, steps_before, names, lambda k: SHAPES[k], layout, s1, s2),
          "param_groups": torch_param_groups(opt_type, 1e-2, len(names))}
    # every hyper-parameter torch writes for this class is there
    assert set(sd["param_groups"][0]) == set(A.state_dict()["param_groups"][0])
    pb = {k: v.detach().clone().requires_grad_(True) for k, v in pa.items()}
    B = cls(list(pb.values()), lr=1e-2)
    B.load_state_dict(sd)
    for k, g in _grads(7).items():
        pa[k].grad = g.clone()
        pb[k].grad = g.clone()
    A.step()
    B.step()                                              # KeyError here before the fix
    for k in names:
        assert torch.equal(pa[k].detach(), pb[k].detach()), (opt_type, k)
