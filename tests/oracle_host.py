"""Adapter exposing the oracle's host-side restatements with the `host` interface of driver.py (tests only)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as orc


class OracleHost:
    def __init__(self, variant="restatement"):
        self.v = variant

    def bal_read(self, path):
        return orc.bal_read(path, self.v)

    def set_prior_lambda(self, *a):
        return orc.set_prior_lambda(*a, variant=self.v)

    def prior_scalings(self, *a):
        return orc.prior_scalings(*a, variant=self.v)

    def slam_create_flags(self, *a):
        return orc.slam_create_flags(*a, variant=self.v)

    def slam_update_flags(self, *a):
        return orc.slam_update_flags(*a, variant=self.v)

    def slam_initialise_new_kf(self, *a):
        return orc.slam_initialise_new_kf(*a, variant=self.v)
