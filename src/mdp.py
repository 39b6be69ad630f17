"""``src.mdp`` of the reference, served by the MI355X engine package."""
from src._leaked import *                            # noqa: F401,F403  (leaked names the notebooks use)
from pomdp_pbvi_exploration_amd.mdp import *         # noqa: F401,F403
from pomdp_pbvi_exploration_amd.mdp import Model, AlphaVector, ValueFunction, VI_Solver, SolverHistory, Solver, log  # noqa: F401
from pomdp_pbvi_exploration_amd.mdp import RewardSet, SimulationHistory, Simulation, Agent   # noqa: F401
