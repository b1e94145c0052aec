import sys, ctypes as C, numpy as np, torch
sys.path.insert(0,'.')
from segmentation_amd import _lib as L, engine as E
lib=L.load()
lib.seg_dbg_set_stamps.argtypes=[C.c_void_p]; lib.seg_dbg_set_stamps.restype=C.c_int
def run(hw,cin,cout,cfg,B=16):
    dt=L.SEG_BF16; dev=torch.device('cuda',0)
    layer=E.Layer('c','conv',3,[cin],cout,'VALID',True)
    store=E.ParamStore([layer],dt,dev,training=True)
    rng=np.random.default_rng(0)
    store.set_params({'c':{'weights':rng.standard_normal(layer.wshape).astype(np.float32)*0.1,'biases':np.zeros(cout,np.float32)}})
    net=E.Net(store,B,dt,dev); s=torch.cuda.current_stream().cuda_stream
    p=E.Plan('pack'); net.pack(p); p.run(s)
    x=net.act(hw,hw,cin); x.t.copy_(torch.randn(x.t.shape,device=dev).to(x.t.dtype)); y=net.act(hw-2,hw-2,cout)
    plan=E.Plan('m'); net.conv_fwd(plan,layer,[(x,0,0)],hw,hw,y,cfg=cfg)
    for _ in range(3): plan.run(s)
    torch.cuda.synchronize()
    nwg=200000
    st=torch.zeros(nwg*4*16,dtype=torch.int64,device=dev)
    assert lib.seg_dbg_set_stamps(st.data_ptr())==0
    plan.run(s); torch.cuda.synchronize()
    lib.seg_dbg_set_stamps(None)
    a=st.cpu().numpy().reshape(-1,16)
    a=a[a[:,0]>0]
    d=np.diff(a[:,:9].astype(np.float64),axis=1)
    names=['prologue(addr)','issue prefetch0','b_addr etc + wait loads -> barrier','commit0','barrier','issue prefetch1','compute0 (72 mfma)','remaining chunks','epilogue']
    print('hw',hw,'cin',cin,'cout',cout,'cfg',cfg,'waves',len(a),'lifetime mean %.0f cycles'%(a[:,8]-a[:,0]).mean())
    for n,v in zip(names,d.mean(0)): print('   %-40s %8.0f'%(n,v))
    t0=a[:,0].min(); t1=a[:,8].max(); print('   kernel span %.0f cycles (100MHz ticks? -> see below)'%(t1-t0))
run(125,64,64,1); run(59,128,128,1); run(38,64,64,2)
