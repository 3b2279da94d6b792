import torch, time
dev="cuda"
N,P=4,65536
for (Cin,Cout) in [(128,256),(256,128)]:
    x=torch.randn(N,Cin,P,device=dev); W=torch.randn(Cout,Cin,device=dev)
    out=torch.empty(N,Cout,P,device=dev)
    for _ in range(3): torch.matmul(W,x,out=out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.matmul(W,x,out=out)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(f"rocBLAS fp32 {Cin}->{Cout}: {ms*1e3:.1f} us {2.0*N*P*Cin*Cout/ms/1e9:.1f} TF  {4.0*N*P*(Cin+Cout)/ms/1e6:.0f} GB/s")
    # wgrad shape: [Cout x NP] x [NP x Cin]
    d=torch.randn(N,Cout,P,device=dev)
    for _ in range(3): torch.einsum('nop,ncp->oc', d, x)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): torch.einsum('nop,ncp->oc', d, x)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"rocBLAS fp32 wgrad {Cout}x{Cin}: {ms*1e3:.1f} us {2.0*N*P*Cin*Cout/ms/1e9:.1f} TF")
