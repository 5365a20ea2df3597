"""Set-up cost of a stereo BA problem at configs[2]: pack, structure (torch vs device builder), torch breakdown."""
import sys,time; sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch
from visual_underwater_slam_amd import synth, ba_pack
s=synth.ba_sequence(2000,50000,1000)
nL=len(s["points_gt"])
dev="cuda:0"
op=torch.from_numpy(s["obs_pose"]).to(dev).long(); ol=torch.from_numpy(s["obs_point"]).to(dev).long(); me=torch.from_numpy(s["meas"]).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t=time.perf_counter()
    pk=ba_pack.pack_observations(op,ol,me,2000,nL); torch.cuda.synchronize(); t1=time.perf_counter()
    st=ba_pack.build_structure(pk); torch.cuda.synchronize(); t2=time.perf_counter()
    from visual_underwater_slam_amd.ba import build_structure_device
    st2=build_structure_device(pk); torch.cuda.synchronize(); t3=time.perf_counter()
    print(f"rep{rep}: pack {1e3*(t1-t):.1f} ms, structure (torch) {1e3*(t2-t1):.1f} ms, structure (csrc/structure.hip) {1e3*(t3-t2):.1f} ms")
    del st, st2
# finer breakdown of build_structure
pkk=pk
import torch
def T(msg, t0):
    torch.cuda.synchronize(); t=time.perf_counter(); print(f"   {msg}: {1e3*(t-t0):.1f} ms"); return t
n_obs,nP=pkk["n_obs"],pkk["n_poses"]
t0=time.perf_counter()
op=pkk["obs_pose"].to(torch.int64); ol=pkk["obs_point"].to(torch.int64); pptr=pkk["point_ptr"].to(torch.int64); ppos=pkk["obs_ppos"].to(torch.int64)
ar=torch.arange(n_obs,device=dev); seg=pptr[ol]; counts=ar-seg+1; total=int(counts.sum().item()); t0=T("prep",t0)
a_idx=torch.repeat_interleave(ar,counts); t0=T("repeat_interleave",t0)
excl=torch.cumsum(counts,0)-counts; b_idx=seg[a_idx]+(torch.arange(total,device=dev)-excl[a_idx]); t0=T("b_idx",t0)
pi,pk_=op[a_idx],op[b_idx]; band=int((pi-pk_).max().item()); key=pi*nP+pk_; t0=T("keys",t0)
ks,order=torch.sort(key,stable=True); t0=T("sort",t0)
uk,cnt=torch.unique_consecutive(ks,return_counts=True); t0=T("unique",t0)
pa=ppos[a_idx[order]].to(torch.int32); pb=ppos[b_idx[order]].to(torch.int32); t0=T("gather pairs",t0)
