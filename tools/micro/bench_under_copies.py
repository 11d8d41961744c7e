"""Does copy-engine traffic slow the kernels down?  The device-only hot path (WelsHipGroupBench) alone, then with a thread that keeps
H2D copies / D2H copies / plain host memcpys running.  python tools/micro/bench_under_copies.py"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench as B
import openh264_amd as oh


class A:
    qp, deblock_idc, host_threads = 24, 0, 32


def main():
    w, h, ring, sessions = 1920, 1080, 8, 256
    fsz = w * h * 3 // 2
    from openh264_amd.utils.synth import synth_sequence
    content = B.Content(synth_sequence(w, h, 2 * ring), fsz, ring, False)      # as bench.py builds it
    g = B.make_group(oh, A, 0, w, h, "p", sessions, ring, content)
    g.bench(1, 0)
    g.bench(10, 0)
    host = torch.empty(800 << 20, dtype=torch.uint8).pin_memory()
    host2 = torch.empty(800 << 20, dtype=torch.uint8)
    dev = torch.empty(800 << 20, dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    for mode in ("alone", "h2d", "d2h", "host_memcpy", "alone"):
        stop = [False]
        count = [0]

        def loop():
            with torch.cuda.stream(st):
                while not stop[0]:
                    if mode == "h2d":
                        dev.copy_(host, non_blocking=True); st.synchronize()
                    elif mode == "d2h":
                        host.copy_(dev, non_blocking=True); st.synchronize()
                    elif mode == "host_memcpy":
                        host2.copy_(host)
                    else:
                        time.sleep(0.01)
                    count[0] += 1
        th = threading.Thread(target=loop)
        th.start()
        time.sleep(0.2)
        t0 = time.perf_counter()
        ev = g.bench(30, 0)
        dt = time.perf_counter() - t0
        stop[0] = True
        th.join()
        print("%-12s md %.2f ms/step, deblock %.2f, total %.2f  (%.0f frames/s; %d x 800 MB copied meanwhile = %.1f GB/s)" % (
            mode, ev["md_ms"] / 30, ev["deblock_ms"] / 30, ev["total_ms"] / 30, sessions * 30 / dt, count[0], count[0] * 0.8388 / dt))
    g.close()


main()
