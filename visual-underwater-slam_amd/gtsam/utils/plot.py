"""Placeholder for gtsam.utils.plot, imported by batch.py:27.  batch.py plots with matplotlib directly
(constr3DPoints / the MSE report are in visual_underwater_slam_amd.report); nothing here is on the hot path."""


def __getattr__(name):
    raise NotImplementedError(f"gtsam.utils.plot.{name} is not provided: plotting is outside the hot path")
