"""``src.pomdp`` of the reference, served by the MI355X engine package."""
import copy                                          # noqa: F401  (leaked names the notebooks use)
from datetime import datetime                        # noqa: F401
import numpy as np                                   # noqa: F401
from pomdp_pbvi_exploration_amd.pomdp import (Model, Belief, BeliefSet, BeliefValueMapping, SolverHistory, Solver, PBVI_Solver,   # noqa: F401
                                              HSVI_Solver, FSVI_Solver, FSVI_EG_Solver, load_POMDP_file,
                                              SimulationHistory, Simulation, SimulationSet, Agent, RewardSet)
from pomdp_pbvi_exploration_amd.mdp import log, ValueFunction, AlphaVector, VI_Solver   # noqa: F401
from pomdp_pbvi_exploration_amd.mdp import Model as MDP_Model   # noqa: F401
