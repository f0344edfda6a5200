import cProfile, pstats, sys, os, io
sys.argv = ['train_probe.py', '4', 'all', 'f32']
sys.path.insert(0, os.path.join(os.getcwd(), 'tools'))
pr = cProfile.Profile()
import runpy
# warm-up run inside the profile is unavoidable (the probe runs 4 steps); profile everything, sort by tottime
pr.enable()
runpy.run_path('tools/train_probe.py', run_name='__main__')
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(35)
print(s.getvalue()[:6000])
