"""The pipelined end-to-end leg of bench.py alone (for rocprofv3 traces): python tools/e2e_pipe_run.py [sessions] [frames] [threads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B          # noqa: E402
import openh264_amd as oh  # noqa: E402


class A:
    qp, deblock_idc, host_threads = 24, 0, 32


def main():
    sessions = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    A.host_threads = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    w, h, ring = 1920, 1080, 8
    fsz = w * h * 3 // 2
    from openh264_amd.utils.synth import synth_sequence
    content = B.Content(synth_sequence(w, h, 2 * ring), fsz, ring, False)      # as bench.py builds it
    t0 = time.perf_counter()
    dt, nbytes, _, host = B.e2e_pipelined_leg(oh, A, 0, w, h, sessions, ring, content, frames, False)
    print("pipelined: %.0f frames/s (%d sessions x %d frames in %.1f ms; second half %.0f frames/s; whole leg %.1f s)" % (
        sessions * frames / dt, sessions, frames, dt * 1e3, B.e2e_pipelined_leg.steady or 0, time.perf_counter() - t0), host)


main()
