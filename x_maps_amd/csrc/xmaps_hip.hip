// xmaps_hip.hip -- host side of libxmaps_hip.so: the C-ABI declared in include/xmaps.h.
// Written for MI355X (gfx950) only: build with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared xmaps_hip.hip -o libxmaps_hip.so
// -ffp-contract=off keeps the time normalisation (divide, multiply, rint) unfused = bit-exact with NumPy.
#include "xmaps_kernels.hpp"
#include "xmaps_k1cols.hpp"
#include "xmaps_k1own.hpp"
#include "xmaps_k2pipe.hpp"
#include "xmaps_ingest.hpp"
#include "xmaps_evt3.hpp"
#include "xmaps_evt2.hpp"

#include <hip/hip_ext.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <limits>
#include <map>
#include <memory>
#include <algorithm>
#include <new>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/xmaps.h"

using namespace xm;

// The host side is one translation unit, split by concern (each file closes the namespaces / linkage blocks it opens):
#include "host/xm_host.hpp"         // errors, slots, workers' queues, the handle, launch macros
#include "host/xm_launch.hpp"       // launch helpers of every kernel variant
#include "host/xm_own_plan.hpp"     // owner-tile tables (host analysis in xm_create)
#include "host/xm_enqueue.hpp"      // path selection + the launches of one frame
#include "host/xm_batch.hpp"        // multi-frame launches (groups)
#include "host/xm_workers.hpp"      // redo of failed shortcuts, launch workers, single-frame entry
#include "host/xm_api_engine.hpp"   // xm_create .. xm_process_batch, adaptive batching
#include "host/xm_api_graph.hpp"    // hipGraph batches
#include "host/xm_api_stage.hpp"    // debug + stage API
#include "host/xm_api_shard.hpp"    // shards (multi-GPU)
#include "host/xm_api_sharded.hpp"  // one frame over several GPUs of one process (RCCL communicators owned by the handle)
#include "host/xm_api_shardcomm.hpp"  // one rank of a frame sharded over several processes: the library drives RCCL itself
#include "host/xm_api_filters.hpp"  // frame event filters, pause detection
#include "host/xm_api_ingest.hpp"   // device-side ingest
#include "host/xm_api_evt3.hpp"     // EVT 3.0 decoder on the device (alone / in front of the ingest)
#include "host/xm_api_misc.hpp"     // X-map builder, evaluation metrics, memory helpers
