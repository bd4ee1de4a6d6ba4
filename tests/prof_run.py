import sys, time
sys.path.insert(0,'.')
import tidehunter_amd as TH
th=TH.TideHunter(device=0, lib_path='tidehunter_amd/libtidehunter_amd_prof.so')
B=int(sys.argv[1]) if len(sys.argv)>1 else 4096
seqs=[TH.synth_read(th.lib,1,i,1000,10,0.1) for i in range(B)]
th.upload(seqs); th.run(); th.collect()
t=time.time(); th.upload(seqs); th.run(); th.collect(); print('time',time.time()-t, th.timing())
