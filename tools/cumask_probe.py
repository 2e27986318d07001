"""Does hipExtStreamCreateWithCUMask hold on gfx950's eight XCDs, and how do mask bits map to (XCD, SE, CU)?
    python tools/cumask_probe.py   (GPU box)"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sigdigger_amd import engine

torch.cuda.init()
ctx = engine.Context(0)
n = ctx.cu_count()
print("compute units:", n)


def summarize(tag, stream, nblocks=4096):
    w = ctx.probe_placement(stream, nblocks, 40000)
    per = collections.Counter(w)
    xcd = collections.Counter(x for x, _, _ in w)
    print(f"{tag}: {len(per)} distinct (xcc, se, cu); workgroups per XCD {[xcd.get(i, 0) for i in range(8)]}")
    return per


summarize("unmasked (current stream)", torch.cuda.current_stream())
for bit in (0, 1, 2, 7, 8, 9, 16, 64, 255):
    st = ctx.masked_stream([bit])
    per = summarize(f"mask = bit {bit}", st, 64)
    print("    ->", sorted(per))
    ctx.destroy_stream(st)
st = ctx.masked_stream(list(range(8)))
per = summarize("mask = bits 0..7", st, 512)
print("    ->", sorted(per))
st2 = ctx.masked_stream(list(range(8, n)))
per2 = summarize(f"mask = bits 8..{n - 1}", st2, 8192)
print("    overlap with bits 0..7:", sorted(set(per) & set(per2)))
# dispatcher round robin over XCDs when the XCDs have different numbers of enabled CUs
st3 = ctx.masked_stream(list(range(4, n)))
summarize("mask = bits 4..255 (XCD 0-3 one CU short)", st3, 8192)
