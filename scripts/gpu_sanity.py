import ctypes as C, torch, time
t0=time.time()
lib = C.CDLL("librecommender_amd/lib/liblibreco_hip.so")
print("loaded", time.time()-t0, torch.cuda.is_available(), torch.cuda.get_device_name(0))
V,K,n=1000,64,5000
tab=torch.randn(V,K,device="cuda"); idx=torch.randint(0,V,(n,),device="cuda",dtype=torch.int32)
out=torch.empty(n,K,device="cuda")
f=lib.lr_embed_gather_f32; f.restype=C.c_int
f.argtypes=[C.c_void_p,C.c_int64,C.c_int,C.c_void_p,C.c_int64,C.c_void_p,C.c_void_p]
rc=f(tab.data_ptr(),V,K,idx.data_ptr(),n,out.data_ptr(),torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("rc",rc,"equal",torch.equal(out,tab[idx.long()]))
# segments
ws_b=lib.lr_segments_ws_bytes; ws_b.restype=C.c_size_t; ws_b.argtypes=[C.c_int64,C.c_int64]
nb=ws_b(n,V); ws=torch.empty(nb,dtype=torch.uint8,device="cuda")
pos=torch.empty(n,dtype=torch.int32,device="cuda"); rows=torch.empty(n,dtype=torch.int32,device="cuda")
start=torch.empty(n+1,dtype=torch.int32,device="cuda"); nseg=torch.zeros(1,dtype=torch.int32,device="cuda")
g=lib.lr_segments_build; g.restype=C.c_int
g.argtypes=[C.c_void_p,C.c_int64,C.c_int64,C.c_void_p,C.c_void_p,C.c_void_p,C.c_void_p,C.c_void_p,C.c_size_t,C.c_void_p]
rc=g(idx.data_ptr(),n,V,pos.data_ptr(),rows.data_ptr(),start.data_ptr(),nseg.data_ptr(),ws.data_ptr(),nb,torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ns=int(nseg.item()); u=torch.unique(idx)
print("seg rc",rc,"nseg",ns,len(u),torch.equal(rows[:ns].long(),u.long()), int(start[ns]))
import subprocess; print(subprocess.run("rocminfo | grep -E 'gfx|Compute Unit' | head -6; nproc; free -g | head -2",shell=True,capture_output=True,text=True).stdout)
