"""Wave cycles per phase of k_analyse_flow (library built with -DPCAMV_PROF, see tools/dbg/README)."""
import ctypes as C, os, subprocess, sys, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.join(R, "video-steganography-pcamv_amd", "libpcamv_gpu_prof.so")
g = sys.argv[1] if len(sys.argv) > 1 else "256"
env = dict(os.environ, PCAMV_GPU_LIB=lib, PCAMV_PROF_DUMP="1")
out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--steps", "2", "--warmup", "1", "--gops", g, "--cpu-frames", "0", "--g-sweep", "", "--clip-keyints", "", "--parity-gops", "0", "--host-io-steps", "0"] + sys.argv[2:], env=env, capture_output=True, text=True)
print(out.stdout[-600:]); print(out.stderr[-2000:])
