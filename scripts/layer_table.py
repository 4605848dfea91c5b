"""Map the conv_gemm dispatches of one bench step (rocprofv3 kernel trace) to ResNet-50 layers."""
import csv, sys
def resnet50_units(N=256):
    units=[("stem",dict(M=N*112*112,Cout=64,K=147))]
    blocks=[]
    cin,h=64,56
    for li,(w,nb) in enumerate(zip((64,128,256,512),(3,4,6,3)),1):
        for b in range(nb):
            s=2 if (b==0 and li>1) else 1
            ho=h//s
            name=f"l{li}.{b}"
            u=[(name+".c1",dict(M=N*h*h,Cout=w,K=cin,Min=N*h*h,Cin=cin)),
               (name+".c2",dict(M=N*ho*ho,Cout=w,K=9*w,Min=N*h*h,Cin=w,s=s)),
               (name+".c3",dict(M=N*ho*ho,Cout=4*w,K=w,Min=N*ho*ho,Cin=w))]
            ds=None
            if s!=1 or cin!=4*w: ds=(name+".ds",dict(M=N*ho*ho,Cout=4*w,K=cin,Min=N*h*h,Cin=cin,s=s))
            blocks.append((u,ds)); cin=4*w; h=ho
    return units,blocks
units,blocks=resnet50_units()
fwd=[units[0]]
for u,ds in blocks:
    if ds: fwd.append(ds)
    fwd+=u
bwd=[]
for u,ds in reversed(blocks):
    for x in reversed(u): bwd.append(x)
    if ds: bwd.append(ds)
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'conv_gemm_kernel' in r['Kernel_Name']][-105:]
assert len(fwd)==53 and len(bwd)==52
tot=0
for (name,d),r in zip(fwd+bwd, rows):
    us=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    fl=2.0*d['M']*d['Cout']*d['K']
    kind='F' if tot<53 else 'D'
    tot+=1
    print(f"{kind} {name:9s} M={d['M']:8d} N={d['Cout'] if kind=='F' else d.get('Cin',3):5d} K={d['K'] if kind=='F' else d['Cout']*(9 if 'c2' in name else 1):5d} {us:8.1f} us {fl/us/1e6:7.1f} TF/s  blocks={int(r['Grid_Size_X'])//256}")
