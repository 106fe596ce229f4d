/* placeholder, filled in later */ int orc_kd_placeholder(void){return 0;}
