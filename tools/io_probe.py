import sys, time, os, json
sys.path.insert(0, ".")
from gms_amd import capi
s = int(sys.argv[1])
t0 = time.time(); csr = capi.HostCSR.generate("kronecker", s); t1 = time.time()
os.makedirs("/tmp/gmsx_cache", exist_ok=True)
p = f"/tmp/gmsx_cache/probe-{s}.sg"
csr.save_sg(p); t2 = time.time()
c2 = capi.HostCSR.load(p, relabel=capi.RELABEL_NEVER); t3 = time.time()
print(json.dumps({"scale": s, "omp": os.environ.get("OMP_NUM_THREADS"), "gen_s": round(t1 - t0, 1), "save_s": round(t2 - t1, 1), "load_s": round(t3 - t2, 1), "bytes": os.path.getsize(p)}))
os.remove(p)
