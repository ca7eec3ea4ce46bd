

def main():
    import sys, time, torch
    sys.path.insert(0, '/root/repo')
    import bench
    dev = torch.device("cuda", 0)
    s, info = bench.build_sampler("abstracts", dev, 0, 1, False)
    for _ in range(50): s.sweep()
    torch.cuda.synchronize()
    for ev in (True, False):
        s.kernel_events = [] if ev else None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3000): s.sweep()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("events", ev, "enqueue us/sweep %.1f  wall us/sweep %.1f" % ((t1 - t0) / 3000 * 1e6, (t2 - t0) / 3000 * 1e6))


if __name__ == "__main__":
    main()
