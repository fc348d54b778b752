#!/usr/bin/env python3
"""Every species of the reference through the executable ON THE DEVICE against the reference binary (which travels with the
repository): the species' own defaults (sample 100 where that is the default), --UTR=off, soft-masking on, --maxDNAPieceSize=30000
(piece cuts on the longer records), five records.  The species-parameter DATA of the reference is not in the repository; pack it
in the build container, untracked, and let gpurun carry it:
    (cd /root/reference && tar -czf /root/repo/tmp_all_config.tar.gz config/species config/model config/extrinsic/extrinsic.cfg \
         config/parameters/aug_cmdln_parameters.json)
    gpurun -- python tests/sweep_species_gpu.py K N SECONDS       (every N-th species from K on, for at most SECONDS)
    gpurun -- python tests/sweep_species_gpu.py K N SECONDS utr   (the species that ship UTR parameters, with --UTR=on: the dense kernels;
                                                                   records without GC-class steps)
Round 2: 148 of 148 species that load byte-identical; the other 19 are refused by name (bacterium gene model, window sizes
outside the trellis kernel's scheduling, missing files of the distribution)."""
import sys, os, subprocess, tarfile, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import *
d=tempfile.mkdtemp()
with tarfile.open("tmp_all_config.tar.gz") as t: t.extractall(d)
cfg=os.path.join(d,"config")+"/"
byname=dict(golden_inputs())
utr=len(sys.argv)>4 and sys.argv[4]=="utr"
recs=[(n,byname[n]) for n in (("HS04636","rand20k_b","trunc_both","revcomp","multigc_levels") if utr else ("HS04636","multigc_levels","softmask_gene","trunc_both","rand20k_b"))]
fa=os.path.join(d,"in.fa"); write_fasta(fa,recs)
env=dict(os.environ,AUGUSTUS_CONFIG_PATH=cfg)
k,nw=int(sys.argv[1]),int(sys.argv[2])
t0=time.time(); nok=nfail=0
for i,sp in enumerate(sorted(os.listdir(cfg+"species"))):
    if i%nw!=k or not os.path.exists(cfg+"species/%s/%s_parameters.cfg"%(sp,sp)): continue
    if time.time()-t0>float(sys.argv[3]): break
    if utr and not any(f.endswith("utr_probs.pbl") for f in os.listdir(cfg+"species/"+sp)): continue
    args=["--species="+sp,"--UTR=on","--softmasking=0",fa] if utr else ["--species="+sp,"--UTR=off","--maxDNAPieceSize=30000",fa]
    ours=subprocess.run(["augustus_amd/bin/augustus"]+args,capture_output=True,text=True,env=env)
    if ours.returncode!=0:
        print(sp,"ours rc",ours.returncode,ours.stderr.strip().splitlines()[-1][:90] if ours.stderr.strip() else "",flush=True); continue
    ref=subprocess.run([REF_AUGUSTUS]+args,capture_output=True,text=True,env=env)
    ok=ref.returncode==0 and gff_body(ref.stdout)==gff_body(ours.stdout)
    nok+=ok; nfail+=(not ok)
    print(sp,"OK" if ok else "FAIL",flush=True)
print("ok",nok,"fail",nfail)
