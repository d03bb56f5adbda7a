/* Source-compatibility forwarder: the reference ships this declaration set as include/polycap-source.h;
 * in this build every public declaration lives in polycap.h. */
#ifndef POLYCAP_FWD_SOURCE_H
#define POLYCAP_FWD_SOURCE_H
#include "polycap.h"
#endif
