import sys, time
sys.path.insert(0, "/root/repo")
import torch
from ocr_amd import engine, opstream
s = opstream.layered_dag(2_000_000, 10_000)
built = s.with_main(s.main[:-1])
prog = engine.StreamProgram(built, device=0, flags=0).build()
eng = prog.eng
eng.snapshot()
for rep in range(3):
    eng.restore()
    eng.event_satisfy_slot(prog._guids(s.main[-1:]["a"]))
    eng.sync()
    t0 = time.perf_counter(); n = 0
    while True:
        ready, pend = eng.step(1)
        n += 1
        if not ready and not pend:
            break
    dt = time.perf_counter() - t0
    print("levels", n, "us per step call", 1e6 * dt / n, flush=True)
eng.restore(); eng.event_satisfy_slot(prog._guids(s.main[-1:]["a"])); eng.sync()
t0 = time.perf_counter(); st = eng.resolve(); dt = time.perf_counter() - t0
print("resolve us per level", 1e6 * dt / st.levels)
