"""What a cross-stream dependency costs on the device: a chain of 200 dependent tiny kernels on ONE stream against the same chain alternating between
two streams (event record + stream wait at every hop), both replayed from a hipGraph (no host in the loop), and from the library's launch tape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = "cuda"
x = torch.zeros(64, device=dev)
N = 200


def chain(two_streams: bool):
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(a):
        with torch.cuda.graph(g, stream=a):
            cur = a
            for i in range(N):
                if two_streams:
                    nxt = b if cur is a else a
                    nxt.wait_stream(cur)
                    cur = nxt
                with torch.cuda.stream(cur):
                    x.add_(1.0)
            a.wait_stream(cur)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(a)
        for _ in range(5):
            g.replay()
        e1.record(a); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * N)


t1, t2 = chain(False), chain(True)
print(f"hipGraph, per kernel: one stream {t1:.2f} us, alternating streams {t2:.2f} us -> {t2 - t1:.2f} us per cross-stream hop")

# the same chain through the library's launch tape (hipLaunchKernel / hipEventRecord / hipStreamWaitEvent per operation from one C loop)
from mgsv_amd import ops_train as tr, tape as _tape
y = torch.zeros(64, device=dev)


def tape_chain(two_streams: bool):
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(a):
        tr.add3(y, y); torch.cuda.synchronize()
        with _tape.LaunchTape.record() as tp:
            cur = a
            for i in range(N):
                if two_streams:
                    nxt = b if cur is a else a
                    nxt.wait_stream(cur)
                    cur = nxt
                with torch.cuda.stream(cur):
                    tr.add3(y, y)
            a.wait_stream(cur)
        torch.cuda.synchronize()
        tp.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(a)
        for _ in range(5):
            tp.replay()
        e1.record(a); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / (5 * N)
        tp.close()
    return t


t3, t4 = tape_chain(False), tape_chain(True)
print(f"launch tape, per kernel: one stream {t3:.2f} us, alternating streams {t4:.2f} us -> {t4 - t3:.2f} us per cross-stream hop")
