import sys, torch
sys.path.insert(0, '/root/repo')
import gvcnn_tf_amd as gv
from gvcnn_tf_amd.training import TrainGVCNN
DEV='cuda:0'
def rel(a,b):
    a,b=a.double().cpu(),b.double().cpu(); return float((a-b).norm()/b.norm())
for backbone,size,N,V in [("inception_v3",171,4,2),("inception_v3",171,16,2),("resnet_v2_50",97,3,2),("resnet_v2_50",129,12,2)]:
    C_,G=5,10
    eng=TrainGVCNN(backbone,N,V,size,size,C_,G,device=DEV)
    P=gv.params.init_backbone_params(eng.plan.param_shapes(),seed=2,perturb_bn=True)
    Hd=gv.params.init_head_params(V,eng.raw.c,eng.final.c,C_,seed=3,spread_scores=True)
    x=(torch.rand(N,V,size,size,3,generator=torch.Generator().manual_seed(0))-0.5).to(DEV)
    labels=torch.randint(0,C_,(N,),generator=torch.Generator().manual_seed(1))
    res={}
    for storage,math in (("f32","bf16x3"),("f32","bf16x1"),("bf16","bf16x3")):
        e=TrainGVCNN(backbone,N,V,size,size,C_,G,backbone_params=P,head_params=Hd,device=DEV,storage=storage,math=math)
        if 'sch' in res: out=e.forward(x,labels,g_scheme=res['sch'][0],g_weight=res['sch'][1])
        else:
            out=e.forward(x,labels); res['sch']=(e.scheme.cpu().numpy(),e.weight.cpu().numpy())
        g={k:v.clone() for k,v in e.backward().items()}
        names=sorted(g)
        flat=torch.cat([g[k].reshape(-1).double().cpu() for k in names])
        res[(storage,math)]=(out[1].float().clone(),out[2].clone(),float(out[3]),flat, e)
    ref=res[("f32","bf16x3")]
    for key in (("f32","bf16x1"),("bf16","bf16x3")):
        r=res[key]
        cos=float((r[3]@ref[3])/(r[3].norm()*ref[3].norm()))
        print(backbone,size,N,key,"S %.4f logits %.4f loss %.5f/%.5f gcos %.4f grel %.3f"%(rel(r[0],ref[0]),rel(r[1],ref[1]),r[2],ref[2],cos,float((r[3]-ref[3]).norm()/ref[3].norm())))
    # per-layer activation error along the net
    e32,e16=ref[4],res[("bf16","bf16x3")][4]
    k=0
    for op in e32.plan.ops:
        if op["kind"]=="bn":
            k+=1
            if k%8==0:
                print("   ",op["name"][-40:],"rel_l2 %.4f"%rel(e16.view(op["y"]).float(),e32.view(op["y"])))
