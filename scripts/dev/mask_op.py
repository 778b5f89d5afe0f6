"""Diagnostic: how long do the torch ops at the head of Behavior_policy.learn take in isolation?"""
import torch, time
E, T1, nA = 32, 91, 5
term = torch.zeros(E, T1, nA, 1, dtype=torch.uint8, device="cuda")
def tm(name, fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:50s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us")
t = term[:, :-1]
tm("term[..., 0].permute(2,0,1).to(float32)", lambda: t[..., 0].permute(2, 0, 1).to(torch.float32))
tm("  ... .contiguous()", lambda: t[..., 0].permute(2, 0, 1).to(torch.float32).contiguous())
tm("term[..., 0].to(float32).permute(2,0,1).contiguous()", lambda: t[..., 0].to(torch.float32).permute(2, 0, 1).contiguous())
m = t[..., 0].permute(2, 0, 1).to(torch.float32).contiguous()
tm("mask.sum(dim=1)", lambda: m.sum(dim=1))
tm("cumsum etc (window sums)", lambda: torch.cat([torch.zeros(nA, 1, device="cuda"), m.sum(dim=1).cumsum(dim=1)], dim=1))
