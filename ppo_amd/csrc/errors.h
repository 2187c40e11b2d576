// Shared error reporting of the translation units of libppo_xp.so (thread-local message behind
// ppo_xp_last_error()).
#pragma once
int ppoxp_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
