"""Why is the overlapped loop body slower inside bench.py's context than stand-alone?  The same loops on contexts that differ in one thing."""
import os, sys, importlib.util
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from better_fastlio2_amd import capi, synth
spec = importlib.util.spec_from_file_location("lb", os.path.join(ROOT, "scripts", "loop_body_step.py")); lb = importlib.util.module_from_spec(spec); spec.loader.exec_module(lb)
world = synth.make_world(5_000_000, 0.2)
for name in sys.argv[1:] or ["plain", "capq100k", "replicate", "torch"]:
    if name == "torch":
        import torch
        torch.zeros(1, device="cuda")
    ctx = capi.LioContext(0.2, max_scan_points=100_000 if name == "capq100k" else 120_000, max_map_points=5_000_000 + 2_000_000)
    ctx.map_build(world.points)
    if name == "replicate":
        ctx.debug_set("replicate_loopback", 1); ctx.replicate_init(b"", 7, 8); ctx.replicate_defer(True)
        ctx.map_build(world.points)
    r = lb.run(ctx, capi, synth, world, 24, staged=False)
    print(name, "serial %.1f us  overlapped %.1f us (p90 %.1f)" % (r["serial"]["median_us"], r["overlapped"]["median_us"], r["overlapped"]["p90_us"]), flush=True)
    ctx.close()
