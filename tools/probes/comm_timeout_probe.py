"""GPU probe: rank 0 of a two-rank communicator whose peer never arrives (GMSX_COMM_TIMEOUT_S=5 python tools/probes/comm_timeout_probe.py)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from gms_amd import capi
capi.init(0)
uid = capi.Comm.unique_id()
t0 = time.time()
print("calling init", flush=True)
try:
    capi.Comm.init(0, 2, uid)
    print("init returned ok", round(time.time()-t0,1), flush=True)
except capi.GmsxError as e:
    print('status', e.status, 'after', round(time.time() - t0, 1), flush=True)
os._exit(0)
