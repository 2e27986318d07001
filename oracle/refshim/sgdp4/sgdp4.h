/* refshim: <sgdp4/sgdp4.h> (libsuscan, absent): orbit_t as printed by Orbit::debug (Suscan/Analyzer.cpp:46-63) */
#ifndef REFSHIM_SGDP4_H
#define REFSHIM_SGDP4_H
#include <sigutils/types.h>
#include <suscan_amd.h>   /* xyz_t */
#ifdef __cplusplus
extern "C" {
#endif
typedef struct orbit {
  char *name;
  int ep_year; double ep_day, rev, drevdt, d2revdt2, bstar, eqinc, ecc, mnan, argp, ascn, smjaxs;
  long norb; int satno;
} orbit_t;
#define orbit_INITIALIZER { NULL, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }
void orbit_finalize(orbit_t *);
int  orbit_init_from_data(orbit_t *, const char *, size_t);
int  orbit_init_from_file(orbit_t *, const char *);
#ifdef __cplusplus
}
#endif
#endif
