// Single translation unit of libbuffer_hip.so (gfx950).  Parts are plain includes so that the
// kernels share the static helpers without relocatable device code.
#include <stdlib.h>
#include "core.hip"
#include "radius.hip"
#include "subsample.hip"
#include "pointops.hip"
#include "vn.hip"
#include "voxelize.hip"
#include "registration.hip"
#include "convnet.hip"
#include "convnet_wg.hip"
#include "convnet_h3.hip"
#include "costnet.hip"
#include "costnet_h3.hip"
#include "preprocess.hip"
