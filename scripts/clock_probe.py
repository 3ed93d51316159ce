import os, sys, time, subprocess, threading
sys.path.insert(0, os.getcwd())
import torch
from srl_amd import hip
hip.require_gpu()
dev="cuda:0"
def sample(tag, stop):
    while not stop.is_set():
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk=[l for l in out.splitlines() if "sclk" in l][:1]; pw=[l for l in out.splitlines() if "Power" in l][:1]
        print(tag, sclk, pw, flush=True)
        time.sleep(0.7)
def run(tag, fn, secs=3.0):
    stop=threading.Event(); t=threading.Thread(target=sample, args=(tag,stop)); 
    for _ in range(5): fn()
    torch.cuda.synchronize(); t.start(); t0=time.time(); n=0
    while time.time()-t0 < secs:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n+=20
    dt=(time.time()-t0)/n; stop.set(); t.join(); print(tag, "ms per call", dt*1e3, flush=True)
M,N,K=16384,512,3136
A=torch.randn(M,K,device=dev); B=torch.randn(N,K,device=dev); C=torch.empty(M,N,device=dev)
run("dense-random FC fwd", lambda: hip.gemm(M,N,K,A.data_ptr(),K,0,B.data_ptr(),K,0,C.data_ptr(),N))
A.zero_(); 
run("zero-A FC fwd", lambda: hip.gemm(M,N,K,A.data_ptr(),K,0,B.data_ptr(),K,0,C.data_ptr(),N))
