// inst_g1.cpp -- the kernel instantiations of group 1 of instances.h (one of the translation units libxrft_hip.so is built from)
#include "gpu_rt.h"
#include "fasty.h"
#include "fasty_iso.h"
#include "fasty_c2c.h"
#include "fastm.h"
namespace xrft {
#define XRFT_KW template __global__
#define XRFT_KI_GROUP 1
#include "instances.h"
}
