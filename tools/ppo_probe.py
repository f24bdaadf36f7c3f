"""Where a PPO update's time goes (GPU box): one synchronised timing per phase of a minibatch step, host enqueue time next to it."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_continuousdoubleauction_amd import ppo  # noqa: E402

dev = torch.device("cuda:0")
B, MB = 1 << 20, 1 << 18
torch.manual_seed(0)
m = ppo.ActorCritic(168).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=5e-5, fused=True)
obs = torch.randn(B, 168, device=dev)
with torch.no_grad():
    acts, logp_old, _ = m.act(obs[:MB])
acts = tuple(a.repeat(4, *([1] * (a.dim() - 1))) for a in acts)
logp_old = logp_old.repeat(4)
adv, ret = torch.randn(B, device=dev), torch.randn(B, device=dev)


def timed(name, fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    print(f"{name:45s} {t * 1e3:8.2f} ms   (host enqueue {t_host * 1e3:7.2f} ms)")


for amp in (True, False):
    print("amp", amp)
    timed("whole update (4 epochs x 4 minibatches)", lambda: ppo.ppo_update(m, opt, obs, acts, logp_old, adv, ret, amp=amp))
    timed("the same samples, one network row per 4 (shared obs)", lambda: ppo.ppo_update(m, opt, obs[:B // 4], acts, logp_old, adv, ret, amp=amp, agents_per_row=4))
x = obs.to(torch.bfloat16)[:MB]
a = tuple(t[:MB].contiguous() for t in acts)
lo, ad, re = logp_old[:MB].contiguous(), adv[:MB].contiguous(), ret[:MB].contiguous()
timed("cast obs to bf16 (1M x 168)", lambda: obs.to(torch.bfloat16))
perm = torch.randperm(B, device=dev)
xb = obs.to(torch.bfloat16)
timed("shuffle gathers of one epoch", lambda: (xb[perm], [t[perm] for t in acts], logp_old[perm], adv[perm], ret[perm]))
timed("trunk forward (no grad)", lambda: torch.no_grad().__enter__() or m.trunk(x))
torch.set_grad_enabled(True)
state = {}


def fwd():
    state["o"], state["v"] = m.trunk(x)
    state["loss"], _ = ppo._FusedPPOLoss.apply(state["o"].float(), state["v"].float(), m.log_std, a[0], a[1], a[2], a[3].float(), lo, ad, re, 0.2, 0.5, 0.01)


timed("forward + fused loss", fwd)


def fb():
    fwd()
    opt.zero_grad(set_to_none=True)
    state["loss"].backward()


timed("forward + loss + backward", fb)
timed("clip_grad_norm_", lambda: torch.nn.utils.clip_grad_norm_(m.parameters(), 0.5, foreach=True))
timed("fused Adam step", lambda: opt.step())
g = torch.randn(MB, 512, device=dev, dtype=torch.bfloat16)
timed("wgrad split-K bmm 168x512", lambda: torch.bmm(g.view(64, MB // 64, -1).transpose(1, 2), x.reshape(64, MB // 64, -1)).sum(0, dtype=torch.float32))
timed("wgrad single GEMM 168x512", lambda: g.t() @ x)
w = torch.randn(512, 168, device=dev, dtype=torch.bfloat16)
timed("fwd GEMM [MB,168]x[168,512]", lambda: x @ w.t())
h = torch.randn(MB, 256, device=dev, dtype=torch.bfloat16)
w2 = torch.randn(256, 256, device=dev, dtype=torch.bfloat16)
timed("fwd GEMM [MB,256]x[256,256]", lambda: h @ w2.t())
hs = g.view(-1, 2, 256).unbind(1)[0]
timed("fwd GEMM strided [MB,256 of 512]x[256,256]", lambda: hs @ w2.t())
timed("tanh [MB,512] bf16", lambda: torch.tanh(g))
