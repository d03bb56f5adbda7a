/* Source-compatibility forwarder: the reference ships this declaration set as include/polycap-photon.h;
 * in this build every public declaration lives in polycap.h. */
#ifndef POLYCAP_FWD_PHOTON_H
#define POLYCAP_FWD_PHOTON_H
#include "polycap.h"
#endif
