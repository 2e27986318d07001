/* refshim: <sigutils/log.h> (absent): the log sink structs of Suscan/Logger.cpp:28-66 */
#ifndef REFSHIM_SIGUTILS_LOG_H
#define REFSHIM_SIGUTILS_LOG_H
#include <sigutils/types.h>
#include <sys/time.h>
#ifdef __cplusplus
extern "C" {
#endif
enum sigutils_log_severity { SU_LOG_SEVERITY_DEBUG, SU_LOG_SEVERITY_INFO, SU_LOG_SEVERITY_WARNING, SU_LOG_SEVERITY_ERROR, SU_LOG_SEVERITY_CRITICAL };
struct sigutils_log_message { enum sigutils_log_severity severity; struct timeval time; const char *domain, *function; unsigned int line; const char *message; };
struct sigutils_log_config { void *priv; SUBOOL exclusive; void (*log_func)(void *privdata, const struct sigutils_log_message *msg); };
void su_log_init(const struct sigutils_log_config *config);
const char *su_log_severity_to_string(enum sigutils_log_severity);
#ifdef __cplusplus
}
#endif
#endif
