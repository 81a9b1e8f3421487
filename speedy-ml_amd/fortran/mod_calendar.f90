! Module mod_calendar of the drop-in: the hybrid's calendar (src/mod_calendar.f90:24-175), integer bookkeeping done by the
! library (sml_calendar_date / sml_hours_into_year reproduce the reference's quirks: 8760-hour years, day 0 = 31 December).
module mod_calendar
  use iso_c_binding
  use mod_utilities, only : calendar_type
  implicit none
  type(calendar_type) :: calendar
  interface
    function sml_calendar_date(startyear, hours_elapsed, date_out) bind(C, name="sml_calendar_date") result(rc)
      import :: c_int
      integer(c_int), value :: startyear, hours_elapsed
      integer(c_int), intent(out) :: date_out(4)
      integer(c_int) :: rc
    end function
    function sml_hours_into_year(year, month, day, hour) bind(C, name="sml_hours_into_year") result(h)
      import :: c_int
      integer(c_int), value :: year, month, day, hour
      integer(c_int) :: h
    end function
  end interface
contains
  subroutine initialize_calendar(datetime, startyear, startmonth, startday, starthour)
    type(calendar_type), intent(inout) :: datetime
    integer, intent(in) :: startyear, startmonth, startday, starthour
    datetime%startyear = startyear; datetime%startmonth = startmonth; datetime%startday = startday; datetime%starthour = starthour
  end subroutine

  subroutine get_current_time_delta_hour(datetime, hours_elapsed)
    type(calendar_type), intent(inout) :: datetime
    integer, intent(in) :: hours_elapsed
    integer(c_int) :: d(4), rc
    rc = sml_calendar_date(int(datetime%startyear, c_int), int(hours_elapsed, c_int), d)
    if (rc < 0) stop 'mod_calendar: sml_calendar_date failed'
    datetime%currentyear = d(1); datetime%currentmonth = d(2); datetime%currentday = d(3); datetime%currenthour = d(4)
  end subroutine

  subroutine numof_hours_into_year(year, month, day, hour, numofhours)
    integer, intent(in) :: year, month, day, hour
    integer, intent(out) :: numofhours
    numofhours = sml_hours_into_year(int(year, c_int), int(month, c_int), int(day, c_int), int(hour, c_int))
  end subroutine

  ! hours of the calendar years startyear .. endyear, leap years counted (src/mod_calendar.f90:94-131)
  subroutine numof_hours(startyear, endyear, numofhours)
    integer, intent(in) :: startyear, endyear
    integer, intent(out) :: numofhours
    integer :: y
    numofhours = 0
    do y = startyear, endyear
      if ((mod(y, 4) == 0 .and. mod(y, 100) /= 0) .or. mod(y, 400) == 0) then
        numofhours = numofhours + 8784
      else
        numofhours = numofhours + 8760
      end if
    end do
  end subroutine

  ! hours from one date to a later one (src/mod_calendar.f90:177-213)
  subroutine time_delta_between_two_dates(start_year, start_month, start_day, start_hour, end_year, end_month, end_day, end_hour, numofhours)
    integer, intent(in) :: start_year, start_month, start_day, start_hour, end_year, end_month, end_day, end_hour
    integer, intent(out) :: numofhours
    integer :: into_start, into_end, between
    call numof_hours_into_year(start_year, start_month, start_day, start_hour, into_start)
    call numof_hours_into_year(end_year, end_month, end_day, end_hour, into_end)
    between = 0
    if (start_year /= end_year) call numof_hours(start_year, end_year - 1, between)
    numofhours = between + into_end - into_start
  end subroutine

  subroutine time_delta_between_two_dates_datetime_type(datatime1, datetime2, timedelta)
    type(calendar_type), intent(inout) :: datatime1, datetime2
    integer, intent(out) :: timedelta
    call time_delta_between_two_dates(datatime1%currentyear, datatime1%currentmonth, datatime1%currentday, datatime1%currenthour, &
                                      datetime2%currentyear, datetime2%currentmonth, datetime2%currentday, datetime2%currenthour, timedelta)
  end subroutine
end module mod_calendar
