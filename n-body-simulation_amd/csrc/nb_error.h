// nb_error.h -- error plumbing shared by the HIP and the plain C++ sources.
// The reference prints and exits on any failure (src/gpu/BH/err.h:21-46); here every failure
// becomes a status code plus a thread-local message read through nb_last_error().
#pragma once

#include <string>

namespace nb {
void set_error(const std::string &msg);
int fail(int status, const std::string &msg);
}  // namespace nb
