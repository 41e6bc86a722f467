#!/bin/bash
# round 6, GPU pass Q: real_dim along the second axis and cross spectra on the fused inner / mid layouts: parity, no copies, timing against the transposing path
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "innermost or fused or not_adjacent or inner_layout" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python - <<'PY' > $O/timing.txt 2>&1
import time, numpy as np, torch, warnings
import xrft_amd as xa
from xrft_amd import api
warnings.simplefilter("ignore")
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): r=fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
def peak(f):
    f(); torch.cuda.synchronize(); b=torch.cuda.memory_allocated(); torch.cuda.reset_peak_memory_stats(); r=f(); torch.cuda.synchronize()
    return torch.cuda.max_memory_allocated()-b, r
for shape,dims,td in (((1024,1024,64),("y","x","t"),["y","x"]),((1460,73,144),("time","lat","lon"),["time","lon"]),((720,91,360),("time","lat","lon"),["time","lon"])):
    x=torch.randn(shape,device="cuda"); y=torch.randn(shape,device="cuda")
    co={d:np.arange(n)*1.0 for d,n in zip(dims,shape)}
    da=xa.DataArray(x,dims,co); db=xa.DataArray(y,dims,co)
    for rd in (None, td[-1]):
        f=lambda: xa.power_spectrum(da,dim=td,real_dim=rd,detrend="linear",window="hann")
        pk,r=peak(f); dt=t(f)
        print(f"PS {shape} dim={td} real_dim={rd}: {x.numel()/dt/1e9:7.1f} GFFT/s  {dt*1e3:7.3f} ms  peak extra {pk/2**20:7.1f} MiB (result {r.data.numel()*r.data.element_size()/2**20:.1f} MiB)  {next(reversed(api._plan_cache.values())).describe().splitlines()[1][:40]}")
    for env in (None, "transposing path"):
        if env:
            inner=api._execute_inner; api._execute_inner=lambda *a,**k: None
        f=lambda: xa.cross_spectrum(da,db,dim=td,detrend="linear",window="hann")
        pk,r=peak(f); dt=t(f)
        print(f"cross_spectrum {shape} dim={td} {env or 'where the axes lie'}: {x.numel()/dt/1e9:7.1f} GFFT/s  {dt*1e3:7.3f} ms  peak extra {pk/2**20:7.1f} MiB (result {r.data.numel()*r.data.element_size()/2**20:.1f} MiB, contiguous {r.data.is_contiguous()})  {next(reversed(api._plan_cache.values())).describe().splitlines()[1][:40]}")
        if env: api._execute_inner=inner
PY
cat $O/timing.txt | grep -v amdgpu
