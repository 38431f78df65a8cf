"""ORACLE (test infrastructure) -- physical constants, GFS_PHYS=True branch of
util/pace/util/constants.py:36-73 (the values the reference's dycore actually uses)."""
RADIUS = 6.3712e6
PI = 3.1415926535897931
OMEGA = 7.2921e-5
GRAV = 9.80665
RGRAV = 1.0 / GRAV
RDGAS = 287.05
RVGAS = 461.50
CP_AIR = 1004.6
KAPPA = RDGAS / CP_AIR
DZ_MIN = 2.0
CV_AIR = CP_AIR - RDGAS
RDG = -RDGAS / GRAV
CNST_0P20 = 0.2
K1K = RDGAS / CV_AIR
ZVIR = RVGAS / RDGAS - 1
HUGE_R = 1.0e40
