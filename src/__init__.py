"""Compatibility package: the reference's notebooks do ``from src.pomdp import *``."""
