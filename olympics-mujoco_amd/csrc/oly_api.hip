// ctx management, error reporting and measurement helpers of libolympic_hip.so.
#include "oly_common.h"

extern "C" const char* oly_version(void) { return "olympic_hip 0.4 (gfx950)"; }
extern "C" int oly_abi_version(void) { return OLY_ABI_VERSION; }

extern "C" const char* oly_strerror(int code) {
  switch (code) {
    case OLY_OK: return "ok";
    case OLY_EINVAL: return "invalid argument";
    case OLY_ENOTCONF: return "entry point used before its configure/upload call";
    case OLY_EHIP: return "HIP runtime error";
    case OLY_ENOMEM: return "out of memory";
    case OLY_ERANGE: return "table entry out of range";
    case OLY_ENODEV: return "no usable gfx950 device";
    default: return "unknown error";
  }
}

extern "C" const char* oly_last_error(const oly_ctx* ctx) { return ctx ? ctx->err : "NULL ctx"; }

extern "C" int oly_create(oly_ctx** out, int device) {
  if (!out) return OLY_EINVAL;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return OLY_ENODEV;
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) return OLY_ENODEV;
  }
  if (device >= count) return OLY_ENODEV;
  oly_ctx* ctx = new (std::nothrow) oly_ctx();
  if (!ctx) return OLY_ENOMEM;
  memset(ctx, 0, sizeof(*ctx));
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
    delete ctx;
    return OLY_ENODEV;
  }
  ctx->num_cu = prop.multiProcessorCount;
  ctx->stats_ws_bytes = sizeof(double) * OLY_STATS_MAX_BLOCKS * 2 * OLY_MAX_OBS;
  if (hipMalloc(&ctx->il_dev, sizeof(IlDev)) != hipSuccess ||
      hipMalloc(&ctx->a3_dev, sizeof(A3Dev)) != hipSuccess ||
      hipMalloc(&ctx->stats_ws, ctx->stats_ws_bytes) != hipSuccess) {
    oly_destroy(ctx);
    return OLY_ENOMEM;
  }
  *out = ctx;
  return OLY_OK;
}

extern "C" void oly_destroy(oly_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->il_dev) (void)hipFree(ctx->il_dev);
  if (ctx->a3_dev) (void)hipFree(ctx->a3_dev);
  if (ctx->stats_ws) (void)hipFree(ctx->stats_ws);
  if (ctx->contact.geom_bodyid) (void)hipFree(ctx->contact.geom_bodyid);
  if (ctx->traj.rows) (void)hipFree(ctx->traj.rows);
  if (ctx->grf.geom_group) (void)hipFree(ctx->grf.geom_group);
  free(ctx->grf_group_host);
  delete ctx;
}

extern "C" int oly_event_create(void** ev) {
  if (!ev) return OLY_EINVAL;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return OLY_EHIP;
  *ev = e;
  return OLY_OK;
}
extern "C" int oly_event_destroy(void* ev) {
  return hipEventDestroy(static_cast<hipEvent_t>(ev)) == hipSuccess ? OLY_OK : OLY_EHIP;
}
extern "C" int oly_event_record(void* ev, oly_stream stream) {
  return hipEventRecord(static_cast<hipEvent_t>(ev), oly_s(stream)) == hipSuccess ? OLY_OK : OLY_EHIP;
}
extern "C" int oly_event_sync(void* ev) {
  return hipEventSynchronize(static_cast<hipEvent_t>(ev)) == hipSuccess ? OLY_OK : OLY_EHIP;
}
extern "C" int oly_event_elapsed_ms(void* start, void* stop, float* ms_out) {
  if (!ms_out) return OLY_EINVAL;
  return hipEventElapsedTime(ms_out, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)) ==
                 hipSuccess
             ? OLY_OK
             : OLY_EHIP;
}
extern "C" int oly_stream_sync(oly_stream stream) {
  return hipStreamSynchronize(oly_s(stream)) == hipSuccess ? OLY_OK : OLY_EHIP;
}
