import os,sys,torch,numpy as np
sys.path.insert(0,'.')
from segmentation_amd.datasets import ArrayDataSet
from segmentation_amd.unet import UNetModel
rng=np.random.default_rng(5555)
x=rng.uniform(0,1,(2,2,188,188,3)).astype(np.float32); y=rng.integers(0,2,(2,2,188,188,1)).astype(np.uint8)
kw=dict(sess=None,n_classes=2,input_dims=188,learning_rate=1e-3,log_dir=None,save_dir=None,load_snapshot=False,dtype='f32')
def run(ws,graph,steps=3):
    m=UNetModel(dataset=ArrayDataSet(x,y),use_graph=graph,wgrad_streams=ws,**kw)
    for _ in range(steps): m.train_step()
    torch.cuda.synchronize()
    return m.store.p.clone(), m.store.g.clone()
ref=run(0,False)
for ws,graph in ((0,False),(0,True),(2,False),(2,True),(2,True),(1,True)):
    p,g=run(ws,graph)
    print('ws',ws,'graph',graph,'p equal',bool(torch.equal(p,ref[0])),'g equal',bool(torch.equal(g,ref[1])),'max|dg|',float((g-ref[1]).abs().max()))
