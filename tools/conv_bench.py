"""fwd+bwd time of the NatureCNN under layouts / dtypes (MIOpen NCHW vs channels-last vs unfold+GEMM, fp32 vs bf16)."""
import torch, time, torch.nn as nn, torch.nn.functional as F
B=4096
def run(name, net, x, n=5):
    opt=torch.optim.Adam(net.parameters(), 1e-4)
    for i in range(n+2):
        if i==2: torch.cuda.synchronize(); t=time.perf_counter()
        y=net(x); loss=y.float().square().mean(); opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter()-t)/n*1e3:.2f} ms fwd+bwd")
def cnn():
    return nn.Sequential(nn.Conv2d(4,32,8,4),nn.ReLU(),nn.Conv2d(32,64,4,2),nn.ReLU(),nn.Conv2d(64,64,3,1),nn.ReLU(),nn.Flatten(),nn.Linear(1024,512),nn.ReLU()).cuda()
class UnfoldConv(nn.Conv2d):
    def forward(self, x):
        B,C,H,W=x.shape; k=self.kernel_size[0]; s=self.stride[0]
        oh=(H-k)//s+1; ow=(W-k)//s+1
        cols=F.unfold(x,k,stride=s)                       # B, C*k*k, L
        y=torch.matmul(self.weight.view(self.out_channels,-1), cols) + self.bias.view(1,-1,1)
        return y.view(B,self.out_channels,oh,ow)
def cnn_unfold():
    return nn.Sequential(UnfoldConv(4,32,8,4),nn.ReLU(),UnfoldConv(32,64,4,2),nn.ReLU(),UnfoldConv(64,64,3,1),nn.ReLU(),nn.Flatten(),nn.Linear(1024,512),nn.ReLU()).cuda()
x=torch.rand(B,4,64,64,device="cuda")
run("nchw fp32", cnn(), x)
run("channels_last fp32", cnn().to(memory_format=torch.channels_last), x.contiguous(memory_format=torch.channels_last))
torch.backends.cudnn.benchmark=True
run("nchw fp32 benchmark", cnn(), x)
run("unfold+gemm fp32", cnn_unfold(), x)
net=cnn()
with torch.autocast("cuda",dtype=torch.bfloat16):
    run("nchw bf16 autocast", net, x)
net=cnn_unfold()
with torch.autocast("cuda",dtype=torch.bfloat16):
    run("unfold bf16 autocast", net, x)
