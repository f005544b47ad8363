// zd_multi.cpp — multi-GPU driver inside the library (filled in below).
#include "zd_plan.h"
