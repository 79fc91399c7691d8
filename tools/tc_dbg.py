import sys, json, os, subprocess
# limiter experiment driver: same graph (cached .sg), one child process per GMSX_DBG value
sys.path.insert(0, ".")
s = int(sys.argv[1])
if len(sys.argv) > 2:
    from gms_amd import capi
    capi.init(0)
    sg = f"/tmp/k{s}.sg"
    if not os.path.exists(sg):
        capi.HostCSR.generate("kronecker", s).save_sg(sg)
    g = capi.DeviceGraph.from_csr(capi.HostCSR.load(sg), flags=capi.UPLOAD_TRUSTED)
    for _ in range(3):
        t, st = g.tc_total(stats=True)
    print(json.dumps({"dbg": os.environ.get("GMSX_DBG"), "tri": t, "kernel_ms": st["kernel_ms"]}), flush=True)
else:
    for d in (0, 16, 4096, 0, 16, 4096):
        subprocess.run([sys.executable, __file__, str(s), "child"], env=dict(os.environ, GMSX_DBG=str(d)))
