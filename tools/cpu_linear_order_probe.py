"""Does this host's torch CPU nn.Linear follow the k-ascending fma chain (see csrc/reference_order.hip)?  Prints the fraction of
mismatching elements per (M, K) - 0.000 means bit-identical."""
import platform, subprocess
import torch, torch.nn.functional as F
def chain(A,W,b):
    acc = torch.zeros(A.shape[0], W.shape[0], dtype=torch.float64)
    for k in range(A.shape[1]): acc = (acc + A[:,k:k+1].double()*W[:,k].double().unsqueeze(0)).float().double()
    return (acc+b.double()).float()
print(torch.__version__, platform.processor()); print(subprocess.run('lscpu | grep -E "Model name|Flags" | cut -c1-400', shell=True, capture_output=True, text=True).stdout)
print(torch.backends.cpu.get_cpu_capability())
g = torch.Generator().manual_seed(0)
for M in (1,3,5,10,17,64,100,500,3001):
    row=[]
    for K,N in ((2,16),(5,40),(16,64),(40,64),(64,64),(64,320),(128,640)):
        A,W,b = 10*torch.randn(M,K,generator=g), 0.3*torch.randn(N,K,generator=g), torch.randn(N,generator=g)
        row.append("%.3f"%((F.linear(A,W,b)!=chain(A,W,b)).float().mean().item()))
    print(M,row)
