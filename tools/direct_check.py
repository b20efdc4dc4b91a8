import sys; sys.path.insert(0, "."); import numpy as np, torch
import __graft_entry__ as e
pkg=e.load_package(); L=pkg.lib(); O=e.load_oracle()
ok=True
for (n,h,w,c) in [(3,16,16,3),(2,1,16,1),(3,9,48,3),(3,33,80,3),(70,256,256,3),(2,1080,1920,3),(2,75,4096,4),(3,131,112,1),(3,17,2064,2),(5,37,320,3),(1,2,262144,1)]:
    host=O.lcg_stream(n,h,w,c)
    for r in (1,2):
        a=torch.from_numpy(host).cuda(); b=torch.full_like(a,0xA5)
        pkg.check(L.mi_blur_enqueue_ex(a.data_ptr(),b.data_ptr(),w,h,c,r,n,0,h,pkg.VARIANT_DIRECT,None)); torch.cuda.synchronize()
        want=O.blur_batch(host,r)
        same=np.array_equal(b.cpu().numpy(),want); ok&=same
        print((n,h,w,c),"r",r,"direct == oracle:",same, "" if same else int((b.cpu().numpy()!=want).sum()))
        if h>=5:
            y0,y1=1,h-2
            b2=torch.full((n,y1-y0,w,c),0xA5,dtype=torch.uint8,device="cuda")
            pkg.check(L.mi_blur_enqueue_ex(a.data_ptr(),b2.data_ptr(),w,h,c,r,n,y0,y1,pkg.VARIANT_DIRECT,None)); torch.cuda.synchronize()
            same=np.array_equal(b2.cpu().numpy(),want[:,y0:y1]); ok&=same
            if not same: print("  band rows",y0,y1,"MISMATCH")
print("ALL OK" if ok else "MISMATCH")
