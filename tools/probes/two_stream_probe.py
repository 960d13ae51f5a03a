#!/usr/bin/env python3
"""GPU box: does running the 8 frames of an FFDNet pass as two half-batches on two HIP streams fill the tail of the
2.67-generation body-layer grids (2048 workgroups on 768 resident slots)?  Ten split-fp16 body layers, one stream vs two."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import ops

n, c, h, w = 8, 96, 256, 256
g = torch.Generator().manual_seed(0)
x = ops.c8_to_c8s(ops.to_c8(torch.randn(n, c, h, w, generator=g).cuda()))
wt = torch.randn(c, c, 3, 3, generator=g) * 0.05
pk = ops.pack_conv3x3_split(wt, None, Cin=c, Cout=c, device='cuda')
bufs = [torch.empty_like(x) for _ in range(2)]


def chain(xin, b0, b1, layers=10):
    cur, src = 0, xin
    for _ in range(layers):
        dst = (b0, b1)[cur]
        ops.conv3x3_c8s(src, pk, c, relu=True, out=dst)
        src = dst
        cur ^= 1


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
halves = [(x[:4], bufs[0][:4], bufs[1][:4]), (x[4:], bufs[0][4:], bufs[1][4:])]


def one():
    chain(x, bufs[0], bufs[1])


def two():
    cur = torch.cuda.current_stream()
    for st, (xi, a, b) in zip((s1, s2), halves):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            chain(xi, a, b)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


for f in (one, two):
    for _ in range(3):
        f()
torch.cuda.synchronize()
for name, f in (('one stream, 8 frames', one), ('two streams, 4 + 4 frames', two), ('one stream, 8 frames', one), ('two streams, 4 + 4 frames', two)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:28s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per 10-layer pass')
