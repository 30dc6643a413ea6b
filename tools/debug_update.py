import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(1, os.path.join(ROOT, "code"))
import models, train
import pytorch_tecogan_amd.train as hip_train
import tecogan_oracle as orc
os.environ["TECOGAN_GRAPH"] = "0"
def synth(B,T,cs,seed):
    rng=np.random.default_rng(seed)
    return (torch.from_numpy(rng.random((B,T,3,cs,cs),dtype=np.float32)), torch.from_numpy(rng.random((B,T,3,4*cs,4*cs),dtype=np.float32)))
x,y=synth(1,10,32,1)
args=orc.default_args(); 
gp=orc.init_params(orc.generator_param_shapes(16),101); dp=orc.init_params(orc.discriminator_param_shapes(4,128),201)
gp0={k:v.clone() for k,v in gp.items()}
bufs=orc.init_bn_buffers(dp); og=orc.AdamState(gp,1e-4); od=orc.AdamState(dp,1e-4)
torch.set_num_threads(16)
net,gg,dg,f=orc.tecogan_step(gp,dp,bufs,og,od,x,y,args,0,return_grads=True)
for chunks in ("1","2"):
    os.environ["TECOGAN_GBWD_CHUNKS"]=chunks
    hip_train._STEPS.clear()
    a=orc.default_args(); a.tg_dtype="fp32"
    G=models.generator(3,a); D=models.discriminator(a)
    G.load_state_dict(gp0); D.load_state_dict(orc.init_params(orc.discriminator_param_shapes(4,128),201),strict=False)
    G,D=G.cuda(),D.cuda()
    o1=torch.optim.Adam(G.parameters(),1e-4,betas=(0.9,0.999),eps=1e-8); o2=torch.optim.Adam(D.parameters(),1e-4,betas=(0.9,0.999),eps=1e-8)
    out=train.FRVSR_Train(x.cuda(),y.cuda(),a,D,G,0,0.,0.,o1,o2); torch.cuda.synchronize()
    for name in ("output.weight","output.bias","conv.0.weight","resids.7.0.weight","conv_trans.6.bias"):
        p=dict(G.named_parameters())[name]
        dw=(p.detach().cpu()-gp[name]); upd_ref=(gp[name]-gp0[name]); upd_hip=(p.detach().cpu()-gp0[name])
        gerr=(p.grad.cpu()-gg[name]).abs().max()/gg[name].abs().max()
        print(f"chunks={chunks} {name:22s} max|w_hip-w_ref|/lr={float(dw.abs().max())/1e-4:.4f}  mean|upd_ref|/lr={float(upd_ref.abs().mean())/1e-4:.3f} mean|upd_hip|/lr={float(upd_hip.abs().mean())/1e-4:.3f} sign-agree={float((torch.sign(upd_ref)==torch.sign(upd_hip)).float().mean()):.4f} grad-relerr={float(gerr):.2e}")
    st=next(iter(hip_train._STEPS.values()))
    print("hyper", st.hyper.cpu().numpy().round(6).tolist())
