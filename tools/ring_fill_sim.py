#!/usr/bin/env python3
"""What a better-filled timeline could buy k_fc_ring at the REAL density ("what the ring's timeline fill is worth"): a host-side model, no GPU.
2500 pairs' (pair, k-slice) units are intervals of 121 ticks on the table's 122 k-tick line (ten overlap at any point).  The product form gives a workgroup eight CONSECUTIVE units
(one per wave); the alternative packs units into tracks by first fit (a wave takes the next unit that starts after its last one ended, G ticks of start-up apart), NW tracks per
workgroup, layers of NW tracks for the overflow, items cut every K units per track.  A tick is priced by the number of waves active in it with the measured table of
profiles/r05_duo_tick_clock.json (0.60 us at one ... 0.89 at eight, +0.1 per wave beyond: the CU's vector-memory path), an all-idle tick at c0 (argv[1], default 0.15 us).
    python tools/ring_fill_sim.py [c0]
Result (c0 = 0.15): consecutive eight 0.78 ms of ring per full-width lock-step (measured: 0.88); first fit needs up to 24 tracks, and with 8 / 10 / 12 waves and items of 3 units per
track it comes to 0.89 / 0.78-0.82 / 0.71 ms (5 per track: 0.86 / 0.75-0.79 / 0.69) -- the overflow layer walks the whole table again at one or two active waves, where a tick costs
as much as a full one.  Twelve compute waves and items of 0.65-1.05 ms for -9 ... -12 % of the ring's CU time: the "+3 ... +6 % on the headline" of section 4c, from the optimistic side."""
import numpy as np, sys
rng=np.random.default_rng(1)
TABLE=250_000_000; P=1_009_058; SEG=2048; L=121
cost_a={0:None,1:.596,2:.67,3:.70,4:.718,5:.816,6:.864,7:.903,8:.886}
def cost(a,c0):
    if a==0: return c0
    if a<=8: return cost_a[a]
    return 0.886+0.1*(a-8)   # TA-bound beyond 8
def units(npairs, nwin, sorted_windows=True, alive=1.0):
    off=np.sort(rng.integers(0,TABLE-P+1,npairs))
    if not sorted_windows: off=rng.permutation(off)
    wins=np.array_split(off,nwin)
    out=[]
    for w in wins:
        keep=rng.random(len(w))<alive
        w=w[keep]
        st=(w[:,None]+np.arange(4)[None,:]*L*SEG).ravel()
        out.append(np.sort(st)/SEG)
    return out
def item_cost(starts_by_track,c0):
    # starts_by_track: list of arrays of start ticks (float); timeline from floor(min) to max+L
    allst=np.concatenate([s for s in starts_by_track if len(s)])
    t0=np.floor(allst.min()); T=int(np.ceil(allst.max()-t0))+L+1
    act=np.zeros(T+1,int)
    for s in allst:
        a=int(s-t0); act[a:a+L]+=1
    c=sum(cost(a,c0) for a in act[:T])
    return c,T
def consecutive(st,NW,c0):
    tot=0;n=0;ticks=0
    for i in range(0,len(st),NW):
        c,T=item_cost([st[i:i+NW]],c0); tot+=c;n+=1;ticks+=T
    return tot,n,ticks
def firstfit(st,NW,K,G,c0):
    # global first-fit into unlimited tracks grouped in layers of NW; cut each layer into items of ~K units per track
    ends=[];assign=[]
    tracks=[]
    for s in st:
        best=-1;be=-1
        for t,e in enumerate(ends):
            if e+G<=s:
                best=t;break      # first fit: lowest track index that is free
        if best<0: ends.append(s+L);tracks.append([s])
        else: ends[best]=s+L;tracks[best].append(s)
    tot=0;n=0;ticks=0
    for l0 in range(0,len(tracks),NW):
        layer=tracks[l0:l0+NW]
        allst=np.sort(np.concatenate([np.array(t) for t in layer]))
        per_item=NW*K
        # cut by start time into chunks with per_item units
        cuts=[allst[i] for i in range(0,len(allst),per_item)]+[np.inf]
        for ci in range(len(cuts)-1):
            lo,hi=cuts[ci],cuts[ci+1]
            sel=[np.array([s for s in t if lo<=s<hi]) for t in layer]
            if sum(len(s) for s in sel)==0: continue
            c,T=item_cost(sel,c0); tot+=c;n+=1;ticks+=T
    return tot,n,ticks,len(tracks)
if __name__=='__main__':
    c0=float(sys.argv[1]) if len(sys.argv)>1 else 0.15
    for alive in (1.0,0.8,0.6):
      for nwin,srt in ((1,True),(4,True),(4,False)):
        W=units(2500,nwin,srt,alive)
        t=n=0
        for st in W:
            c,k,_=consecutive(st,8,c0);t+=c;n+=k
        print(f'alive {alive} nwin {nwin} sorted {srt}: consecutive-8 CU-ms {t/1000:.1f} items {n} per item {t/n:.1f} -> {t/256/1000:.3f} ms')
        if srt:
          for NW in (8,10,12):
            for K in (2,3,5):
                t=n=0;nt=0
                for st in W:
                    c,k,_,ntr=firstfit(st,NW,K,3,c0);t+=c;n+=k;nt=max(nt,ntr)
                print(f'   firstfit NW {NW} K {K}: CU-ms {t/1000:.1f} items {n} per item {t/n:.1f} tracks {nt} -> {t/256/1000:.3f} ms')
