// inst_g6.cpp -- the kernel instantiations of group 7 of instances.h (fastn.h; one of the translation units libxrft_hip.so is built from)
#include "gpu_rt.h"
#include "fasty.h"
#include "fastm.h"
#include "fastn.h"
namespace xrft {
#define XRFT_KW template __global__
#define XRFT_KI_GROUP 7
#include "instances.h"
}
