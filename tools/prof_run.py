import sys, time
sys.path.insert(0, '.')
import tidehunter_amd as TH
lib = sys.argv[2] if len(sys.argv) > 2 else None
th=TH.TideHunter(device=0, lib_path=lib)
B=int(sys.argv[1]) if len(sys.argv)>1 else 4096
seqs=[TH.synth_read(th.lib,1,i,1000,10,0.1) for i in range(B)]
th.upload(seqs); th.run(); th.collect_native()
t0=time.time(); th.upload(seqs); t1=time.time(); th.run(); t2=time.time(); th.collect_native(); t3=time.time(); r=th.records(); t4=time.time()
print('upload %.3f run %.3f collect_c %.3f records_py %.3f' % (t1-t0,t2-t1,t3-t2,t4-t3), {k:round(v,1) for k,v in th.timing().items()})
