"""Module-level names the reference's ``src/mdp.py`` / ``src/pomdp.py`` leak through ``from src.pomdp import *`` and
that its notebooks use without importing them (``plt``, ``pd``, ``copy``, ``COLOR_ARRAY`` ...; reference
``src/mdp.py:1-37``, ``src/pomdp.py:1-21``).  Plotting itself is out of scope for the engine; these are the library
objects themselves, imported if installed.  CuPy is not shimmed: ``cp`` exists only where CuPy does."""
import copy                                          # noqa: F401
import os                                            # noqa: F401
import random                                        # noqa: F401
from datetime import datetime                        # noqa: F401
from inspect import signature                        # noqa: F401
from typing import Tuple, Union                      # noqa: F401

import numpy as np                                   # noqa: F401

try:
    import pandas as pd                              # noqa: F401
except ImportError:                                  # pragma: no cover
    pass
try:
    from tqdm.auto import trange                     # noqa: F401
except ImportError:                                  # pragma: no cover
    pass
try:
    from matplotlib import animation, cm, colors, patches, ticker     # noqa: F401
    from matplotlib import pyplot as plt             # noqa: F401
    from matplotlib.lines import Line2D              # noqa: F401
    from matplotlib.patches import Rectangle         # noqa: F401
    # the reference's colour tables (src/mdp.py:30-37): the ten Tableau colours as name / id / hex / rgb
    COLOR_LIST = [{'name': item.replace('tab:', ''), 'id': item, 'hex': value,
                   'rgb': [int(value.lstrip('#')[i:i + 2], 16) for i in (0, 2, 4)]}
                  for item, value in colors.TABLEAU_COLORS.items()]
    COLOR_ARRAY = np.array([c['rgb'] for c in COLOR_LIST])
except ImportError:                                  # pragma: no cover
    pass
try:
    from scipy.optimize import LinearConstraint, milp     # noqa: F401
    ilp_support = True
except ImportError:                                  # pragma: no cover
    ilp_support = False

gpu_support = True      # the HIP engine is the GPU backend; a missing library surfaces on first use (never a CPU fallback)
