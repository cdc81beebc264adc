"""Wall time per advance() of the 2D solver at the reference's native 128x192, GPU vs the CPU oracle:
the first steps after a reset (Poisson solver at its iteration cap) and the warm-started steady state."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fsamd_loader
fs = fsamd_loader.load()
from oracle.oracle2d import Oracle2D

def run(make, adv, iters, label):
    g = make()
    out = []
    for phase, n in (("first 10", 10), ("next 40", 40), ("steady 50", 50)):
        t = time.perf_counter()
        its = 0
        for _ in range(n):
            adv(g)
            its += iters(g)
        out.append(f"{phase}: {(time.perf_counter() - t) / n * 1e3:.3f} ms/step ({its / n:.0f} iters)")
    print(label, "; ".join(out), flush=True)

def gpu(ordering):
    def make():
        g = fs.FluidSim2D(128, 192)
        g.params(ordering=ordering)
        return g
    return make

run(gpu(fs.ORDER_LEXICOGRAPHIC), lambda g: g.advance(), lambda g: g.info()["last_iters"], "GPU lexicographic")
run(gpu(fs.ORDER_RED_BLACK), lambda g: g.advance(), lambda g: g.info()["last_iters"], "GPU red-black    ")
run(lambda: Oracle2D(128, 192), lambda o: o.advance(), lambda o: o.last_iters, "CPU oracle (lexi)")
