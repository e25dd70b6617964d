// BatchNorm constants shared by the kernels (nn.BatchNorm2d defaults, ava/models/vae.py:135-141,162-168).
// (A finalisation fused into the producing kernel -- last-arriving workgroup, grouped tickets, or pulled into the
// consumer -- was built three ways and measured slower than the 5 us finalisation launches every time; DESIGN.md
// section 3 has the numbers.  The code was removed.)
#pragma once
#include "common.h"

#define AVA_BN_EPS 1e-5
#define AVA_BN_MOMENTUM 0.1
