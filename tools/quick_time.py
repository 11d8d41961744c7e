import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openh264_amd as oh
from openh264_amd.utils.synth import synth_sequence
w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
yuv = synth_sequence(w, h, 2)
fsz = w * h * 3 // 2
enc = oh.Encoder()
p = enc.GetDefaultParams()
p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.uiIntraPeriod, p.fMaxFrameRate = w, h, 24, 1, 30.0
assert enc.InitializeExt(p) == 0, enc.last_error()
print(enc.backend_name())
for i in range(2):
    enc.EncodeFrame(yuv[(i % 2) * fsz:(i % 2 + 1) * fsz])
t = time.time()
tot = 0
for i in range(n):
    rc, ft, bs, nals = enc.EncodeFrame(yuv[(i % 2) * fsz:(i % 2 + 1) * fsz])
    tot += len(bs)
dt = time.time() - t
print("frames %d  %.2f ms/frame  %.1f fps  bytes %d" % (n, dt / n * 1000, n / dt, tot))
