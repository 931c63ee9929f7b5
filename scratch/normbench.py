"""Device time of F.normalize's pass over one Reddit-sized row (114 M bf16), alone on the chip.
usage: normbench.py <lib.so>   (BLISS_NORM_WGS = workgroups per row)"""
import ctypes as C, os, sys, torch
lib = C.CDLL(os.path.abspath(sys.argv[1]))
P, I64 = C.c_void_p, C.c_int64
lib.bliss_exp3_normalize_global.argtypes = [P, I64, P, P, P, P, P]
dev = torch.device("cuda:0")
E = 113988365
w = (torch.rand(E, device=dev) * 1e-8 + 4e-9).bfloat16()
row_sum = torch.zeros(96, dtype=torch.int64, device=dev)
scratch = torch.zeros(98, dtype=torch.int64, device=dev)
out_norm = torch.zeros(1, dtype=torch.bfloat16, device=dev)
def limbs_for(norm):
    v = int(norm * 2.0 ** 64); l = torch.zeros(96, dtype=torch.int64)
    l[0], l[1], l[2] = v & 0xffffffff, (v >> 32) & 0xffffffff, v >> 64
    return l.to(dev)
la, lb = limbs_for(1.0078125), limbs_for(0.99609375)
st = torch.cuda.current_stream().cuda_stream
def run(l):
    assert lib.bliss_exp3_normalize_global(w.data_ptr(), E, row_sum.data_ptr(), l.data_ptr(), scratch.data_ptr(), out_norm.data_ptr(), st) == 0
for _ in range(3): run(la); run(lb)
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
for i in range(20):
    evs[i].record(); run(la if i % 2 == 0 else lb)
evs[20].record(); torch.cuda.synchronize()
ts = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(20))
print("%s wgs=%s pass us: min %.1f median %.1f max %.1f  -> %.2f TB/s at the median (4 B per edge)" % (
    os.path.basename(sys.argv[1]), os.environ.get("BLISS_NORM_WGS", "1024"), ts[0], ts[10], ts[-1], 4 * E / ts[10] / 1e6))
