// stand-in for CUDA's <math_constants.h>: CUDART_PI is a DOUBLE constant there as well
#pragma once
#define CUDART_PI 3.1415926535897931e+0
