import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import synth, miekki_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for i in range(n):
    ix = miekki_amd.Miekki(31, 12, 8, 33, 10)
    ix.insert_sequences([synth.genome_bases(i, 0, 20000)])
    ix.query([synth.genome_bases(i, 100, 1000)], 10, 1, 0.0)
    ix.close()
maps = open("/proc/self/maps").read()
libs = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libhsa-runtime" in l})
print("before torch:", libs)
import torch
maps = open("/proc/self/maps").read()
libs = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libhsa-runtime" in l})
print("after import torch:", libs)
try:
    x = torch.zeros(4, device="cuda"); print("torch cuda ok", x.sum().item())
except Exception as e:
    print("torch cuda FAILED:", e)
