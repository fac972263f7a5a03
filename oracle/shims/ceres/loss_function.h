// oracle/_ref shim: see ceres.h
#include "ceres.h"
