# does a host-to-device DMA stream slow the decomposition kernel down?  The bench batch resident in HBM, its kernel timed alone and while a side
# thread copies N MB per launch from pinned host memory into a scratch device buffer (torch, its own stream):   python tools/h2d_interference.py
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, aletsch_amd as A
pg = A.synth(seed=1002, n_graphs=100000, v_min=64, v_max=64, fixed_edges=256)
dev = torch.device("cuda:0")
with A.DecompBatch(0) as b:
    b.add(pg); b.upload()
    for _ in range(2): b.run(); b.download()
    for mb, direction in ((0, "h2d"), (340, "h2d"), (675, "h2d"), (1350, "h2d"), (2700, "h2d"), (1350, "d2h"), (1350, "d2d"), (0, "h2d")):
        stop = False; copied = [0]
        def pump():
            if mb == 0: return
            torch.cuda.set_device(dev)
            n = mb * (1 << 20)
            host = torch.empty(n, dtype=torch.uint8).pin_memory(); devb = torch.empty(n, dtype=torch.uint8, device=dev); dev2 = torch.empty(n, dtype=torch.uint8, device=dev) if direction == "d2d" else None
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                while not stop:
                    t0 = time.perf_counter()
                    if direction == "h2d": devb.copy_(host, non_blocking=True)
                    elif direction == "d2h": host.copy_(devb, non_blocking=True)
                    else: dev2.copy_(devb, non_blocking=True)
                    s.synchronize(); copied[0] += 1
                    dt = time.perf_counter() - t0
                    if dt < 0.040: time.sleep(0.040 - dt)          # one copy of `mb` per ~40 ms: the rate of the staged pipeline
        th = threading.Thread(target=pump); th.start()
        time.sleep(1.0 if mb else 0.0)
        ms = []
        for _ in range(8): b.run(); b.download(); ms.append(b.kernel_ms())
        stop = True; th.join()
        print("side copy %5d MB per 40 ms (%s): kernel ms min %.2f median %.2f max %.2f   (copies made: %d)" % (mb, direction, min(ms), sorted(ms)[len(ms) // 2], max(ms), copied[0]), flush=True)
