"""MI355X-native PBVI alpha-vector backup engine (HIP kernels behind a C-ABI) with the
host-side mirror of the reference's Model / ValueFunction / PBVI_Solver interface."""
from . import mdp, pomdp, synth            # noqa: F401
from .pomdp import (Model, Belief, BeliefSet, BeliefValueMapping, PBVI_Solver, FSVI_Solver, FSVI_EG_Solver, HSVI_Solver,   # noqa: F401
                    SolverHistory, load_POMDP_file)
from .mdp import AlphaVector, ValueFunction, VI_Solver, log, set_quiet   # noqa: F401

__all__ = ['Model', 'Belief', 'BeliefSet', 'BeliefValueMapping', 'PBVI_Solver', 'FSVI_Solver', 'FSVI_EG_Solver', 'HSVI_Solver',
           'SolverHistory', 'load_POMDP_file', 'AlphaVector', 'ValueFunction', 'VI_Solver', 'log', 'set_quiet']
