"""Reporting helpers with the reference's own formulas: trajectory
read-back `constr3DPoints` (/root/reference/batch.py:57-68) and the MSE against odometry with the
hard-coded z offset (/root/reference/batch.py:362-367).  Host-side numpy post-processing; the DOT writer
is NonlinearFactorGraph.saveGraph."""
import numpy as np

from .gtsam.symbol_shorthand import X

ODOM_Z_OFFSET = 0.7433      # batch.py:363


def constr3DPoints(values):
    """Positions of X(0), X(1), ... while they exist.  Like the reference, row 0 of the result is an
    uninitialised placeholder row (`np.empty((1, 3))`, batch.py:59) and the poses start at row 1."""
    i = 0
    points = np.empty((1, 3))
    while values.exists(X(i)):
        pose_i = values.atPose3(X(i))
        points = np.append(points, [np.array([pose_i.x(), pose_i.y(), pose_i.z()])], axis=0)
        i += 1
    return points


def trajectory_mse(points, odom_xyz):
    """batch.py:362-366: mean squared difference between the odometry positions (z shifted by 0.7433)
    and the optimised positions points[1:]."""
    odom = np.array(odom_xyz, dtype=float, copy=True).reshape(-1, 3)
    odom -= np.array([0, 0, ODOM_Z_OFFSET])
    return float(np.mean(np.square(odom - np.asarray(points)[1:, :])))
