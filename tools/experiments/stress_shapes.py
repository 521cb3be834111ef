import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import cases, cpu_ref
from metrabs_amd import kernels
from metrabs_amd.config import MetrabsConfig
def mcfg(c): return MetrabsConfig.from_any(c.as_dict())
g=cases.gen(1)
# decode J=555
for (B,J,D,H,W) in [(3,555,8,8,8),(2,555,8,12,12),(1,1000,4,4,4)]:
    cfg=cpu_ref.HeadConfig(depth=D, proc_side=H*32)
    x=torch.randn(B,J*(1+D),H,W,generator=g)*3
    o2,o3=cpu_ref.heads_from_logits(x,J,cfg)
    c2,c3=kernels.softargmax_decode(x.cuda(),J,mcfg(cfg))
    print('decode',(B,J,D,H,W),float((c3.cpu()-o3).abs().max()))
# fused head J=555 / J=122 12x12 / 16x16
for (B,C,J,H) in [(4,1280,555,8),(3,512,122,12),(2,256,30,16),(70,96,17,8)]:
    cfg=cpu_ref.HeadConfig(proc_side=H*32)
    feat=torch.randn(B,C,H,H,generator=g)
    w,b=cases.default_conv_init(J*9,C,g); w,b=w*3,b*3
    o2,o3=cpu_ref.heads_forward(feat,w,b,J,cfg)
    packed=kernels.head_pack_weights(w.cuda(),b.cuda(),J,8,torch.float32)
    c2,c3=kernels.head_fused(feat.cuda(),packed,C,J,mcfg(cfg))
    print('head',(B,C,J,H),float((c3.cpu()-o3).abs().max()))
# reconstruct J=555
for (B,J) in [(64,555),(300,122),(1,1),(2000,17)]:
    cfg=cpu_ref.HeadConfig()
    c2=torch.rand(B,J,2,generator=g)*256; rel=torch.randn(B,J,3,generator=g)*300
    K=cases.intrinsics_for(256,256)[None].repeat(B,1,1) if hasattr(cases,'intrinsics_for') else None
    K=torch.tensor([[500.,0,128],[0,500,128],[0,0,1]])[None].repeat(B,1,1)
    o=cpu_ref.reconstruct_absolute(c2,rel,K,cfg)
    r=kernels.reconstruct_absolute(c2.cuda(),rel.cuda(),K.cuda(),mcfg(cfg))
    print('recon',(B,J),cpu_ref.mpjpe(r.cpu(),o))
