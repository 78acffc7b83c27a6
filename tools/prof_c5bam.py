import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
import torch
import bench
from instrain_amd import engine
from tests import util
torch.cuda.set_device(0)
bench.bind_to_gpu_numa_node(torch, 0)
ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
import instrain_amd.profile.profile_utilities as pu
orig = pu.profile_bam
calls = []
def wrapped(*a, **k):
    pr = cProfile.Profile()
    pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()
        calls.append(pr)
import instrain_amd.profile as amd
amd.profile_bam = wrapped
r = bench.c5_bam_leg(ctx, 16)
print(r["seconds"], r["stages_ms"])
s = io.StringIO()
pstats.Stats(calls[-1], stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
