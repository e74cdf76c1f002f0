// Stand-in (see README.md): data/landmark.h includes sqlite3.h for its (de)serialisation members, which the fixtures never call.
#ifndef SVGPU_SHIM_SQLITE3_H
#define SVGPU_SHIM_SQLITE3_H
typedef struct sqlite3 sqlite3;
typedef struct sqlite3_stmt sqlite3_stmt;
typedef long long sqlite3_int64;
#define SQLITE_OK 0
#define SQLITE_ERROR 1
#define SQLITE_TRANSIENT ((void (*)(void*)) - 1)
static inline int sqlite3_bind_blob(sqlite3_stmt*, int, const void*, int, void (*)(void*)) { return SQLITE_ERROR; }
static inline int sqlite3_bind_int64(sqlite3_stmt*, int, sqlite3_int64) { return SQLITE_ERROR; }
static inline const void* sqlite3_column_blob(sqlite3_stmt*, int) { return nullptr; }
static inline int sqlite3_column_bytes(sqlite3_stmt*, int) { return 0; }
static inline sqlite3_int64 sqlite3_column_int64(sqlite3_stmt*, int) { return 0; }
static inline const char* sqlite3_errmsg(sqlite3*) { return "sqlite3 stand-in"; }
#endif
