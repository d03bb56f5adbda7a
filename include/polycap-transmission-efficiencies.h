/* Source-compatibility forwarder: the reference ships this declaration set as include/polycap-transmission-efficiencies.h;
 * in this build every public declaration lives in polycap.h. */
#ifndef POLYCAP_FWD_TRANSMISSION_EFFICIENCIES_H
#define POLYCAP_FWD_TRANSMISSION_EFFICIENCIES_H
#include "polycap.h"
#endif
