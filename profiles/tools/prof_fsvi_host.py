import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
sys.argv = ['olfactory_fsvi.py', '--expansions', '300', '--growth', '100', '--dtype', os.environ.get('PROF_DTYPE', 'f32')]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('examples/olfactory_fsvi.py', run_name='__main__')
finally:
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
    print(s.getvalue()[:9000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25)
    print(s.getvalue()[:6000])
