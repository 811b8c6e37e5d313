// svt_error.h -- thread-local error text shared by every translation unit of libsvtyper_hip.so
#ifndef SVT_ERROR_H
#define SVT_ERROR_H

#include <string>

namespace svt {

inline thread_local std::string g_err;   // returned by svt_last_error()

inline int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

}  // namespace svt

#endif  // SVT_ERROR_H
