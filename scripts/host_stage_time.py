"""Host stages of one stream (the calling thread's work per chunk): PPMd, fxcm's text parser -- us/byte."""
import ctypes as C, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from cmix_amd import engine as E, synth, pipeline
L = E.lib()
data = np.frombuffer(pipeline.text_file_stream(synth.enwik_like(1 << 17, 1000)), np.uint8)
n = len(data)
vocab = np.zeros(256, np.uint8); vocab[np.unique(data)] = 1
p = E.Ppmd(vocab, memory_mb=2000)
for lo in (0, n // 2):
    t = time.time(); p.run(data[lo:lo + n // 2]); print("ppmd        %.2f us/byte" % ((time.time() - t) / (n // 2) * 1e6))
L.fxp_create.restype = C.c_void_p; L.fxp_create.argtypes = [C.c_char_p]
L.fxp_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
fp = L.fxp_create(None)
out = np.zeros(n * 424, np.uint8)
t = time.time(); L.fxp_run(fp, data.ctypes.data, n, out.ctypes.data); print("fxcm parser %.2f us/byte" % ((time.time() - t) / n * 1e6))
# paq8's front end: see host_us_per_byte.paq8_front_enqueue in the bench line (cmx_pipeline_host_ms)
