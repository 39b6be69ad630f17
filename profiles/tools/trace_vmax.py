import sys, os, runpy, time
import numpy as np
sys.path.insert(0, os.getcwd())
from pomdp_pbvi_exploration_amd import engine as E
orig = E.Engine._max_value_objects
calls = [0]
def patched(self, alpha_objects, belief_objects, alpha_values, belief_values, alpha_owner, belief_owner, exact):
    calls[0] += 1
    a_ids = self.row_ids('alpha', alpha_objects, alpha_values, alpha_owner)
    aset = np.unique(a_ids)
    pool = [e for e in self._vmax_cache if e['exact'] == exact]
    desc = []
    for e in pool:
        missing = np.setdiff1d(e['aset'], aset)
        desc.append((len(e['aset']), len(missing), missing[:5].tolist(), int(np.isnan(e['vals']).sum()), len(e['vals'])))
    sub = any(d[1] == 0 for d in desc)
    if not sub and len(aset) > 3000:
        print('call', calls[0], 'V', len(aset), 'dups in a_ids', len(a_ids) - len(aset), 'B', len(belief_objects), 'cache', desc)
    return orig(self, alpha_objects, belief_objects, alpha_values, belief_values, alpha_owner, belief_owner, exact)
E.Engine._max_value_objects = patched
sys.argv = ['olfactory_fsvi.py', '--expansions', '300', '--growth', '100', '--dtype', 'f64']
runpy.run_path('examples/olfactory_fsvi.py', run_name='__main__')
