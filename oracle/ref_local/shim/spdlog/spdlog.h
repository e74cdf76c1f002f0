// Stand-in (see ../README.md): the extractor only uses a trace macro.
#ifndef SVGPU_SHIM_SPDLOG_H
#define SVGPU_SHIM_SPDLOG_H
#define SPDLOG_TRACE(...) (void)0
#define SPDLOG_DEBUG(...) (void)0
namespace spdlog {
template <class... A>
inline void trace(A&&...) {}
template <class... A>
inline void debug(A&&...) {}
template <class... A>
inline void info(A&&...) {}
template <class... A>
inline void warn(A&&...) {}
template <class... A>
inline void error(A&&...) {}
}  // namespace spdlog
#endif
