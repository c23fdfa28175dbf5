import torch, sys
sys.path.insert(0,'/root/repo')
import bench
E,H1,W1,L,r=2,48,64,4,3
vols,coords,offs=bench.make_inputs(E,H1,W1,L,r,1234,'cpu',from_fmaps=False)
rd=7
d=torch.arange(-r,r+1)
di=d.view(1,1,1,rd,1); dj=d.view(1,1,1,1,rd)
def lines(layout):
    # layout: (gx, gy, th, tw): group of gx*gy pixels shares a line of th x tw window tiles; returns bytes per unit
    gx,gy,th,tw=layout
    tot=0.0
    for l in range(L):
        H2,W2=H1>>l,W1>>l
        x0=(coords[:,0]/2**l).view(E,H1,W1,1,1); y0=(coords[:,1]/2**l).view(E,H1,W1,1,1)
        if offs[l] is not None:
            o=offs[l].clone(); o[:,:,:,r,r]=0
            ox,oy=o[...,0]+x0,o[...,1]+y0
        else:
            ox,oy=x0.expand(E,H1,W1,rd,rd),y0.expand(E,H1,W1,rd,rd)
        x1=torch.floor(ox).long()+di; y1=torch.floor(oy).long()+dj
        valid=(x1>=0)&(x1<W2)&(y1>=0)&(y1<H2)
        ids=[]
        ntx=(W2+tw-1)//tw
        for ddy in (0,1):
            for ddx in (0,1):
                xx,yy=x1+ddx,y1+ddy
                ok=valid&(xx<W2)&(yy<H2)
                tid=(yy//th)*ntx+(xx//tw)
                ids.append(torch.where(ok,tid,torch.full_like(tid,-1)))
        ids=torch.stack(ids,-1).view(E,H1,W1,-1)   # per pixel tile ids
        # group pixels
        ids=ids.view(E,H1//gy,gy,W1//gx,gx,-1).permute(0,1,3,2,4,5).reshape(E,H1//gy,W1//gx,-1)
        ids,_=torch.sort(ids,-1)
        uniq=(ids[...,1:]!=ids[...,:-1]).sum(-1)+1-(ids[...,0]<0).long()
        tot+=uniq.float().mean().item()*128/(gx*gy)
    return tot
for lay in [(1,1,4,8),(1,1,8,4),(2,1,4,4),(1,2,4,4),(2,1,2,8),(2,2,2,4),(2,2,4,2),(4,1,2,4),(4,1,4,2),(2,2,2,4),(4,2,2,2),(4,4,2,1),(8,1,2,2),(1,1,2,16)]:
    print(lay, round(lines(lay),1))
