// Stand-in (see ../README.md): the extractor only uses a trace macro.
#ifndef SVGPU_SHIM_SPDLOG_H
#define SVGPU_SHIM_SPDLOG_H
#define SPDLOG_TRACE(...) (void)0
#define SPDLOG_DEBUG(...) (void)0
#endif
