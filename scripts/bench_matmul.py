import torch, time
dev='cuda'
def bench(f, flops, name, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/reps
    print(f'{name}: {ms:.3f} ms {flops/ms/1e9:.1f} TF/s',flush=True)
n,P=160,4096
dy=torch.randn(n,128,P,device=dev); col=torch.randn(n,1152,P,device=dev); W=torch.randn(128,1152,device=dev)
fl=2.0*n*P*128*1152
bench(lambda: torch.bmm(dy, col.transpose(1,2)), fl, 'dW bmm [n,128,P]x[n,P,1152]')
bench(lambda: torch.bmm(dy, col.transpose(1,2)).sum(0), fl, 'dW bmm+sum')
bench(lambda: torch.matmul(W.t(), dy), fl, 'dcol matmul W^T[1152,128] x dy[n,128,P]')
out=torch.empty(n,1152,P,device=dev)
bench(lambda: torch.matmul(W.t(), dy, out=out), fl, 'dcol matmul out=')
# single big-K GEMM with contiguous K
dy2=dy.transpose(0,1).reshape(128,n*P).contiguous(); col2=col.transpose(0,1).reshape(1152,n*P).contiguous()
bench(lambda: dy2 @ col2.t(), fl, 'dW single GEMM K=n*P')
torch.backends.cuda.matmul.allow_tf32=False
print(torch.backends.cuda.preferred_blas_library())
