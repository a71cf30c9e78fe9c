"""Per-phase averages of the stamps written by prof.exe (see run.sh).  Phases: the `each` calls of band_program in
order; `B1/B2/B3` = time spent in the workgroup barriers of wide streams; ticks = shader cycles."""
import numpy as np, sys
def load(p): return np.fromfile(p, dtype=np.uint64).astype(np.int64)
def wide_labels(n_sub_passes):
    mids = ["mid_c+swap"]*(n_sub_passes-2)   # sub-FFT of 512 points: the exchange before its last pass stays in registers
    inv = ["inv0","scat0"]+mids+["sub_last_inv","B2"]
    fb = ["B1","sub_first","scat0"]+mids+["zsplit_c","zsplit_w","mask"]+inv
    # interior workgroups (the recorded one): the loop is rotated, tail_c + head end the trip
    return fb+["tail_lr+head","head_w"]+fb+["tail_lr","B3","stage_c"]+inv+["tail_c+head","head_w"]
def plain_labels(n_passes, wave_sync, swap=False):
    """swap: the exchange before the last pass stays in registers (N = 512, 1024): one phase instead of two"""
    def e2(a,b): return [a,b] if wave_sync else [a,a+"|bar",b,b+"|bar"]
    def e1(a): return [a] if wave_sync else [a,a+"|bar"]
    mids=[]
    for i in range(n_passes-2): mids+=(e1("mid_c+swap") if swap else e2("mid_c","mid_w"))
    inv = e2("inv0","scat0")+mids
    fb = mids+e2("zsplit_c","zsplit_w")+e1("mask")+inv
    return fb+e2("tail_lr+head","head_w")+fb+e2("tail_lr","stage_c")+inv+e2("tail_c+head","head_w")   # rotated (interior)
def report(path, labels, pro, iters):
    t=load(path); per=len(labels)
    assert len(t)>=pro+per*iters, (len(t),pro,per,iters)
    d=np.diff(t)
    acc=np.zeros(per); cnt=0
    for it in range(2,iters-1):
        base=pro+it*per-1   # delta ending at stamp (pro+it*per+k) is d[index-1]
        acc+=d[base:base+per]; cnt+=1
    acc/=cnt
    tot=acc.sum()
    print(path, "ticks/iteration %.0f"%tot, " total span", t[-1]-t[0])
    agg={}
    for l,v in zip(labels,acc):
        print("   %-16s %8.0f  %5.1f%%"%(l,v,100*v/tot))
        agg[l]=agg.get(l,0)+v
    print("  aggregated:")
    for l,v in sorted(agg.items(), key=lambda kv:-kv[1]): print("   %-16s %8.0f  %5.1f%%"%(l,v,100*v/tot))
if __name__=="__main__":
  which=sys.argv[1]
  if which=="all":
      report("gpurun_out/prof/p_13_1_w0.bin", wide_labels(3), 5, 28)
      report("gpurun_out/prof/p_13_1_w7.bin", wide_labels(3), 5, 28)
      report("gpurun_out/prof/p_12_1_w0.bin", wide_labels(2), 5, 14)
      report("gpurun_out/prof/p_10_0_w0.bin", plain_labels(3,True,True), 4, 14)
      report("gpurun_out/prof/p_8_0_w0.bin", plain_labels(2,True), 4, 14)
  if which=="13w": report("gpurun_out/prof/p_13_1_w%s.bin"%sys.argv[2], wide_labels(3), 5, 28)
  if which=="12w": report("gpurun_out/prof/p_12_1_w0.bin", wide_labels(2), 5, 14)
  if which=="13p": report("gpurun_out/prof/p_13_0_w0.bin", plain_labels(4,False), 4, 28)
  if which=="12p": report("gpurun_out/prof/p_12_0_w0.bin", plain_labels(3,False), 4, 14)
  if which=="10p": report("gpurun_out/prof/p_10_0_w0.bin", plain_labels(3,True,True), 4, 14)
  if which=="8p": report("gpurun_out/prof/p_8_0_w0.bin", plain_labels(2,True), 4, 14)
