"""Cost of each piece of the step under hipGraph replay, alone and in lane combinations (the profiler serialises concurrent
streams, so overlap can only be measured by wall time): captures each piece of TecoGANStep as its own graph and times
  * every piece alone,
  * lane A (prep, chain, G backward) alone, lane B (D real, D fake) alone,
  * the whole schedule (TecoGANStep._run_lanes) with the current TECOGAN_* switches."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import models as M, train as TR
import bench as B


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    args = B.default_args("bf16")
    torch.manual_seed(1)
    dev = torch.device("cuda", 0)
    G, D = M.generator(3, args).to(dev), M.discriminator(args).to(dev)
    og = torch.optim.Adam(G.parameters(), 1e-4); od = torch.optim.Adam(D.parameters(), 1e-4)
    x, y = B.synth(4, 10, 32, 1); x, y = x.to(dev), y.to(dev)
    os.environ["TECOGAN_GRAPH"] = "1"
    for s in range(3):
        TR.FRVSR_Train(x, y, args, D, G, s, 0., 0., og, od)
    torch.cuda.synchronize()
    st = next(iter(TR._STEPS.values()))
    g = st.graphs
    print(f"lanes={st.lanes} cu_reserve={st.reserve} dreal_bwd_early={st.dreal_bwd_early}")
    if not st.lanes:
        print(f"{'forward+backward (one forked graph)':44s} {timeit(g[0]):7.3f} ms")
        print(f"{'update':44s} {timeit(g[1]):7.3f} ms")
        return
    for k in st.PIECES:  # buffers hold valid data from the warm-up steps, so every piece can replay alone
        if k in st.LANE_B:
            def on_b(k=k):
                with torch.cuda.stream(st.sBm if k in ("prep", "d_real") else st.sB):
                    g[k]()
            print(f"{k + ' alone (lane B stream)':44s} {timeit(on_b):7.3f} ms")
        else:
            print(f"{k + ' alone':44s} {timeit(g[k]):7.3f} ms")

    def lane_a():
        g["chain0"](); g["chain"](); g["chain_tail"](); g["g_bwd"]()

    def lane_b():
        with torch.cuda.stream(st.sBm):
            g["prep"](); g["d_real"]()
        if st.sBm is not st.sB:
            st.ev["dreal"].record(st.sBm); st.sB.wait_event(st.ev["dreal"])
        with torch.cuda.stream(st.sB):
            g["d_fake"](); g["d_fake_bwd"]()

    def chain_and_dreal():
        main = torch.cuda.current_stream()
        st.ev["start"].record(main); st.sBm.wait_event(st.ev["start"])
        with torch.cuda.stream(st.sBm):
            g["prep"](); st.ev["prep"].record(st.sBm); g["d_real"]()
        g["chain0"](); main.wait_event(st.ev["prep"]); g["chain"]()

    if st.sA is not None:
        torch.cuda.set_stream(st.sA)  # lane A's own stream (lane B's masked stream serialises against the default stream)
    print(f"{'lane A alone (chain, G backward)':44s} {timeit(lane_a):7.3f} ms")
    print(f"{'lane B alone (prep, D real, D fake)':44s} {timeit(lane_b):7.3f} ms")
    print(f"{'chain || prep + D real':44s} {timeit(chain_and_dreal):7.3f} ms")
    print(f"{'whole step (both lanes + update)':44s} {timeit(lambda: st._run_lanes(g)):7.3f} ms")


if __name__ == "__main__":
    main()
