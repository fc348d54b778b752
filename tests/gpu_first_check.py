# first GPU smoke: GPU vs twin, bitwise
import os, sys, time
os.environ['AUGX_DEBUG_CELLS']='1'
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import augustus_amd as ax
from helpers import *
cfg=config_path()
m=ax.Model(cfg,'human')
S=m.n_states
print('states',S, ax.lib().augx_version())
d=ax.Decoder(m,0)
recs=read_fasta(os.path.join(GOLDEN,'example.fa'))
seqs=[s for _,s in recs]+[random_dna(30000,1),random_dna(5000,2).lower(),'N'*3000,random_dna(100,3)]
b=ax.Batch(d,seqs)
t0=time.time(); b.decode(); t1=time.time()
print('decode wall',t1-t0,b.kernel_ms())
res=b.paths()
ok=True
for i,(s,r) in enumerate(zip(seqs,res)):
    rc,lnv,path,V,gc=twin_decode(m.tables_ptr,s,S,cells=True)
    same_path = r.states==path
    cells_ok='-'
    if set(s)!={'N'}:
        Vg=b.cells(i); cells_ok=bool(((Vg==V)|(np.isnan(Vg)&np.isnan(V))).all())
    print(i,len(s),'status',r.status,'lnv gpu',r.ln_viterbi,'twin',lnv,'eq',r.ln_viterbi==lnv,'path',same_path,'cells',cells_ok)
    ok &= same_path and r.ln_viterbi==lnv and cells_ok in (True,'-')
print('ALL OK' if ok else 'MISMATCH')
b.close()
# timing: 16 x 200kb
os.environ['AUGX_DEBUG_CELLS']='0'
d2=ax.Decoder(m,0)
seqs=[random_dna(200000,100+i) for i in range(16)]
b=ax.Batch(d2,seqs)
for it in range(2):
    t0=time.time(); b.decode(); t1=time.time()
    print('16x200kb decode wall',t1-t0,'Mbp/s',3.2/(t1-t0),b.kernel_ms())
res=b.paths()
rc,lnv,path,V,gc=twin_decode(m.tables_ptr,seqs[0],S)
print('200kb check', res[0].ln_viterbi==lnv, res[0].states==path)
