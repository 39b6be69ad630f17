"""``src.pomdp`` of the reference, served by the MI355X engine package."""
from src._leaked import *                            # noqa: F401,F403  (leaked names the notebooks use)
from pomdp_pbvi_exploration_amd.pomdp import (Model, Belief, BeliefSet, BeliefValueMapping, SolverHistory, Solver, PBVI_Solver,   # noqa: F401
                                              HSVI_Solver, FSVI_Solver, FSVI_EG_Solver, load_POMDP_file,
                                              SimulationHistory, Simulation, SimulationSet, Agent, RewardSet)
from pomdp_pbvi_exploration_amd.mdp import log, ValueFunction, AlphaVector, VI_Solver   # noqa: F401
from pomdp_pbvi_exploration_amd.mdp import Model as MDP_Model   # noqa: F401
from pomdp_pbvi_exploration_amd.mdp import SimulationHistory as MDP_SimulationHistory   # noqa: F401
from pomdp_pbvi_exploration_amd.mdp import SolverHistory as MDP_SolverHistory           # noqa: F401
from pomdp_pbvi_exploration_amd.mdp import Solver as MDP_Solver                         # noqa: F401
from pomdp_pbvi_exploration_amd.mdp import Simulation as MDP_Simulation                 # noqa: F401
