// atan2_check_dd.cpp -- TEST INFRASTRUCTURE: cs_atan2() without its fast path (see atan2_check.cpp).
#define CS_ATAN2_NO_FAST 1
#include "../cube_slam_wu_amd/csrc/cs_atan2.h"
extern "C" double atan2_check_dd(double y, double x) { return cs::cs_atan2(y, x); }
